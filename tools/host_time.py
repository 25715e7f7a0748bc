#!/usr/bin/env python3
"""dev tool: where the HOST time of a small step goes (c1 / c2 @ 1k are host-bound: kernels 41 us, step 82 us).
Wall time of each piece of the reference benchmark's step without synchronising in between (the GPU queue never fills
at these sizes), then a cProfile of the whole step.  Usage: host_time.py [workload] [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msda_triton_amd import multiscale_deformable_attention, synth  # noqa: E402

wl = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c1_readme"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
d = synth.make_inputs_torch(wl, "cuda", seed=1)
v, l, a = (d[k].requires_grad_(True) for k in ("value", "loc", "attn"))
sh = d["shapes"]


def step():
    out = multiscale_deformable_attention(v, sh, l, a, wl.padding_mode, wl.align_corners)
    out.backward(torch.rand_like(out))
    v.grad = l.grad = a.grad = None


for _ in range(200):
    step()
torch.cuda.synchronize()
t = {"fwd": 0.0, "rand_like": 0.0, "backward": 0.0, "clear": 0.0}
t0 = time.perf_counter()
for _ in range(n):
    a0 = time.perf_counter()
    out = multiscale_deformable_attention(v, sh, l, a, wl.padding_mode, wl.align_corners)
    a1 = time.perf_counter()
    g = torch.rand_like(out)
    a2 = time.perf_counter()
    out.backward(g)
    a3 = time.perf_counter()
    v.grad = l.grad = a.grad = None
    a4 = time.perf_counter()
    t["fwd"] += a1 - a0
    t["rand_like"] += a2 - a1
    t["backward"] += a3 - a2
    t["clear"] += a4 - a3
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"{wl.name}: step {total / n * 1e6:.1f} us wall;", ", ".join(f"{k} {x / n * 1e6:.1f}" for k, x in t.items()))
with torch.no_grad():
    t0 = time.perf_counter()
    for _ in range(n):
        multiscale_deformable_attention(v, sh, l, a, wl.padding_mode, wl.align_corners)
    print(f"  forward under no_grad: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call")
    torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(1000):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
