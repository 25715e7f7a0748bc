"""Host-side cost of one fwd+bwd step on a small problem (c1 / c4): enqueue rate vs GPU rate, and the CPU-op breakdown.
Usage: python tools/host_overhead.py [workload] [steps] [option=int ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msda_triton_amd import synth
from msda_triton_amd.functional import multiscale_deformable_attention

from msda_triton_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "c1_readme"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
for kv in sys.argv[3:]:
    _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
wl = synth.WORKLOADS[name.split(":")[0]]
if ":" in name:  # e.g. c2_q10k:Q=2000
    import dataclasses
    wl = dataclasses.replace(wl, **{kv.split("=")[0]: int(kv.split("=")[1]) for kv in name.split(":")[1:]})
prof_on = os.environ.get("HOST_PROFILE", "1") != "0"
dev = torch.device("cuda", 0)
d = synth.make_inputs_torch(wl, dev, seed=0)
img, shapes = d["value"].requires_grad_(True), d["shapes"]
pts, attn = d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)
go = None

def step():
    out = multiscale_deformable_attention(img, shapes, pts, attn, wl.padding_mode, wl.align_corners)
    out.backward(go if go is not None else torch.rand_like(out))
    img.grad = pts.grad = attn.grad = None

for _ in range(50): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{name}: enqueue {t_enq/steps*1e6:.1f} us/step, with final sync {t_all/steps*1e6:.1f} us/step")
go = torch.rand_like(multiscale_deformable_attention(img, shapes, pts, attn, wl.padding_mode, wl.align_corners))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"{name} (fixed grad_out): enqueue {t_enq/steps*1e6:.1f} us/step, with final sync {t_all/steps*1e6:.1f} us/step")
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): multiscale_deformable_attention(img, shapes, pts, attn, wl.padding_mode, wl.align_corners)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"{name} fwd only: enqueue {t_enq/steps*1e6:.1f} us, with final sync {t_all/steps*1e6:.1f} us")
if not prof_on:
    sys.exit(0)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(200): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
