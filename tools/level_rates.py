#!/usr/bin/env python3
"""Dev tool: forward / bwd_sample kernel time per sample for pyramids made of fine or coarse levels only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msda_triton_amd.functional import KernelTimer, msda_hip_fwd, msda_hip_bwd
dev = "cuda:0"
B, Q, H, D, P = 4, 10000, 8, 32, 4
for name, levels in (("all four", [(64, 64), (32, 32), (16, 16), (8, 8)]), ("64+32", [(64, 64), (32, 32)]),
                     ("16+8", [(16, 16), (8, 8)]), ("64 only", [(64, 64)]), ("8 only", [(8, 8)]),
                     ("128 only", [(128, 128)])):
    L, I = len(levels), sum(h * w for h, w in levels)
    torch.manual_seed(0)
    v = torch.randn(B, I, H, D, device=dev)
    s = torch.tensor(levels, device=dev)
    l = torch.rand(B, Q, H, L, P, 2, device=dev)
    a = torch.softmax(torch.randn(B, Q, H, L, P, device=dev), -1)
    g = torch.rand(B, Q, H, D, device=dev)
    for _ in range(3):
        msda_hip_fwd(v, s, l, a, "border", True); msda_hip_bwd(g, v, s, l, a, "border", True, (False, True, True))
    torch.cuda.synchronize()
    with KernelTimer() as kt:
        for _ in range(10):
            msda_hip_fwd(v, s, l, a, "border", True); msda_hip_bwd(g, v, s, l, a, "border", True, (False, True, True))
        torch.cuda.synchronize()
    res = {k: ms * 1e3 for k, (n, ms) in kt.summary().items()}
    ns = B * Q * H * L * P
    print(f"{name:10s} samples {ns/1e6:5.2f}M  fwd {res['msda_fwd']:6.1f} us = {res['msda_fwd']*1e6/ns:5.1f} ps/sample   "
          f"bwd_sample {res['msda_bwd_sample']:6.1f} us = {res['msda_bwd_sample']*1e6/ns:5.1f} ps/sample")
