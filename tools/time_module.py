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

# forward + backward of the module core (gradients w.r.t. value, the projection and the reference points)
value.requires_grad_(True)
proj.requires_grad_(True)
ref.requires_grad_(True)
g = torch.rand(B, Q, H, D, device=dev)


def fb_fused():
    out = fused_module_core(value, s, proj, ref, "border", True)
    out.backward(g)
    value.grad = proj.grad = ref.grad = None


def fb_unfused():
    pts, att = module_sampling_inputs(proj, s, ref)
    out = multiscale_deformable_attention(value, s, pts, att, "border", True)
    out.backward(g)
    value.grad = proj.grad = ref.grad = None


def fb_op_only():
    out = multiscale_deformable_attention(value, s, pts0, att0, "border", True)
    out.backward(g)
    value.grad = pts0.grad = att0.grad = None


with torch.no_grad():
    pts0, att0 = module_sampling_inputs(proj, s, ref)
pts0.requires_grad_(True)
att0.requires_grad_(True)
print(f"module core fwd+bwd: fused fwd {bench(fb_fused)*1e3:.1f} us, unfused {bench(fb_unfused)*1e3:.1f} us, "
      f"operator alone {bench(fb_op_only)*1e3:.1f} us")

# kernel-level: the fused backward's sample kernel vs the plain one (grad_value off in both)
from msda_triton_amd.functional import msda_hip_bwd, msda_hip_bwd_fused
with torch.no_grad():
    v_, pr_, rf_ = value.detach(), proj.detach(), ref.detach()
    p0, a0 = pts0.detach(), att0.detach()
    t_plain = bench(lambda: msda_hip_bwd(g, v_, s, p0, a0, "border", True, (False, True, True)))
    t_fused = bench(lambda: msda_hip_bwd_fused(g, v_, s, pr_, rf_, "border", True, need_img=False))
    t_plain_all = bench(lambda: msda_hip_bwd(g, v_, s, p0, a0, "border", True, (True, True, True)))
    t_fused_all = bench(lambda: msda_hip_bwd_fused(g, v_, s, pr_, rf_, "border", True, need_img=True))
print(f"bwd sample-only: plain {t_plain*1e3:.1f} us, fused {t_fused*1e3:.1f} us;  full bwd: plain {t_plain_all*1e3:.1f} us, "
      f"fused {t_fused_all*1e3:.1f} us")
