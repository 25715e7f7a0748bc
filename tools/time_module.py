#!/usr/bin/env python3
"""Dev tool: module core forward, fused prologue vs PyTorch prologue + operator (c2-10k sized)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msda_triton_amd import multiscale_deformable_attention
from msda_triton_amd.functional import fused_module_core, module_sampling_inputs

dev = "cuda:0"
B, Q, H, D, P = 4, 10000, 8, 32, 4
levels = [(64, 64), (32, 32), (16, 16), (8, 8)]
L, I = len(levels), sum(h * w for h, w in levels)
torch.manual_seed(0)
value = torch.randn(B, I, H, D, device=dev)
proj = torch.randn(B, Q, H, L, P, 3, device=dev)
ref = torch.rand(B, Q, 2, device=dev)
s = torch.tensor(levels, device=dev)


def bench(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


with torch.no_grad():
    t_fused = bench(lambda: fused_module_core(value, s, proj, ref, "border", True))
    def unfused():
        pts, att = module_sampling_inputs(proj, s, ref)
        return multiscale_deformable_attention(value, s, pts, att, "border", True)
    t_unfused = bench(unfused)
    err = (fused_module_core(value, s, proj, ref, "border", True) - unfused()).abs().max().item()
print(f"module core forward @ B=4 Q=10k: fused {t_fused*1e3:.1f} us, unfused {t_unfused*1e3:.1f} us, max abs diff {err:.2e}")
