#!/usr/bin/env python3
"""Dev tool: where the host time of one eager fwd+bwd step goes (cProfile, c1-sized problem)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msda_triton_amd import synth, multiscale_deformable_attention
dev = "cuda:0"
wl = synth.WORKLOADS["c1_readme"]
d = synth.make_inputs_torch(wl, dev, seed=0)
v, l, a = (d[k].requires_grad_(True) for k in ("value", "loc", "attn"))
s, g = d["shapes"], d["grad_out"]
def step():
    out = multiscale_deformable_attention(v, s, l, a, wl.padding_mode, wl.align_corners)
    out.backward(g)
    v.grad = l.grad = a.grad = None
for _ in range(50): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): step()
torch.cuda.synchronize()
print("wall per step: %.1f us" % ((time.perf_counter() - t0) / 500 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
