#!/usr/bin/env python3
"""Dev experiment: grad_value of c2-10k as one call vs two half-batch calls on two streams (do the place pass
(L2-write-bound) and the gather (latency-bound) of different halves overlap usefully?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msda_triton_amd import synth
from msda_triton_amd.functional import msda_hip_bwd

dev = "cuda:0"
wl = synth.WORKLOADS["c2_q10k"]
torch.manual_seed(0)
v = torch.randn(wl.B, wl.I, wl.H, wl.D, device=dev)
s = torch.tensor(wl.levels, device=dev)
l = torch.rand(wl.B, wl.Q, wl.H, wl.L, wl.P, 2, device=dev)
a = torch.softmax(torch.randn(wl.B, wl.Q, wl.H, wl.L, wl.P, device=dev), -1)
g = torch.rand(wl.B, wl.Q, wl.H, wl.D, device=dev)
pm, ac = wl.padding_mode, wl.align_corners
needs = (True, False, False)
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
streams = [torch.cuda.Stream() for _ in range(parts)]
chunks = [(i * wl.B // parts, (i + 1) * wl.B // parts) for i in range(parts)]

def one():
    msda_hip_bwd(g, v, s, l, a, pm, ac, needs)

def split():
    cur = torch.cuda.current_stream()
    for st, (b0, b1) in zip(streams, chunks):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            msda_hip_bwd(g[b0:b1], v[b0:b1], s, l[b0:b1], a[b0:b1], pm, ac, needs)
    for st in streams:
        cur.wait_stream(st)

def seq():
    for (b0, b1) in chunks:
        msda_hip_bwd(g[b0:b1], v[b0:b1], s, l[b0:b1], a[b0:b1], pm, ac, needs)

for name, fn in (("one call", one), (f"{parts} streams", split), (f"{parts} sequential", seq), ("one call", one), (f"{parts} streams", split)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name:16s} {(time.perf_counter() - t0) / n * 1e6:8.1f} us per backward (grad_value only)")
