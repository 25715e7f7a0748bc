"""dev tool: the headline step (c2 @ 10k, fwd + bwd) timed step by step from a cold start — how long a fresh process takes to
reach its steady-state step time (GPU clocks, allocator, first launches); what a K = 20 / W = 5 window sees against K = 50 /
W = 10.   python tools/step_ramp.py [idle_ms]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msda_triton_amd import multiscale_deformable_attention, synth

dev = torch.device("cuda", 0)
wl = synth.WORKLOADS["c2_q10k"]
d = synth.make_inputs_torch(wl, dev, seed=0)
img, shapes = d["value"].requires_grad_(True), d["shapes"]
pts, attn = d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)


def step():
    out = multiscale_deformable_attention(img, shapes, pts, attn, wl.padding_mode, wl.align_corners)
    out.backward(torch.rand_like(out))
    img.grad = pts.grad = attn.grad = None


idle = float(sys.argv[1]) / 1e3 if len(sys.argv) > 1 else 0.0
n = 120
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
torch.cuda.synchronize()
time.sleep(idle)
t0 = time.perf_counter()
ev[0].record()
for i in range(n):
    step()
    ev[i + 1].record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("first steps (ms):", " ".join("%.3f" % x for x in ms[:12]))
for a, b in ((0, 5), (5, 25), (10, 60), (60, 120)):
    print("steps [%3d, %3d): mean %.4f ms" % (a, b, sum(ms[a:b]) / (b - a)))
print("wall per step over all %d: %.4f ms" % (n, wall / n * 1e3))
# the bench's own recipe: W untimed steps, barrier, K timed steps by the host clock
for W, K in ((5, 20), (10, 50), (5, 20), (10, 50)):
    torch.cuda.synchronize()
    time.sleep(idle)
    for _ in range(W):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    print("W=%d K=%d: %.4f ms per step" % (W, K, (time.perf_counter() - t0) / K * 1e3))
