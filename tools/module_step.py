"""dev tool: N training steps of the MultiscaleDeformableAttention module at the c2 shape (B=4, Q=10 000, emb = hidden =
256, H=8, L=4, P=4) under bf16 autocast — run it under rocprofv3 --kernel-trace --stats to list every kernel of the step
(tools/prof_module_step.sh), or alone for the step time (after 200 ms of untimed steps: a GPU that has just idled is 6-10 %
slower).   python tools/module_step.py [fp32|bf16] [value_dtype=bf16] [steps] [opt:key=int ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msda_triton_amd import MultiscaleDeformableAttention, synth

dev = torch.device("cuda", 0)
wl = synth.WORKLOADS["c2_q10k"]
shapes = torch.tensor(wl.levels, device=dev)
EMB = wl.H * wl.D
autocast = "fp32" not in sys.argv[1:]
vdt = torch.bfloat16 if any(a.startswith("value_dtype") for a in sys.argv[1:]) else None
steps = next((int(a) for a in sys.argv[1:] if a.isdigit()), 30)
torch.manual_seed(0)
m = MultiscaleDeformableAttention(EMB, EMB, wl.L, wl.H, wl.P, wl.padding_mode, wl.align_corners, value_dtype=vdt).to(dev)
img = torch.randn(wl.B, wl.I, EMB, device=dev, requires_grad=True)
q = torch.randn(wl.B, wl.Q, EMB, device=dev, requires_grad=True)
ref = torch.rand(wl.B, wl.Q, 2, device=dev)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        out = m(img, shapes, q, ref)
    out.float().sum().backward()
    m.zero_grad(set_to_none=True)
    img.grad = q.grad = None


from msda_triton_amd import _lib  # noqa: E402

for a in sys.argv[1:]:
    if a.startswith("opt:"):
        k, v = a[4:].split("=")
        _lib.set_option(k, int(v))
for _ in range(8):
    step()
torch.cuda.synchronize()
t_spin = time.perf_counter()
while time.perf_counter() - t_spin < 0.2:
    for _ in range(10):
        step()
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print("module step (%s autocast, value_dtype=%s): %.4f ms" % ("bf16" if autocast else "no", vdt, (time.perf_counter() - t0) * 1e3 / steps))
