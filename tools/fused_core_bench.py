"""dev tool: per-kernel times (the library's own event pairs) of the module CORE — fused_module_core forward + backward at
the c2 module shape — for option strings alternated inside one process, next to the unfused operator on the same
inputs:   python tools/fused_core_bench.py [value=bf16 | storage] [ref4] [--] opt=val[,opt=val] ...      ("-" = defaults;
"storage": value, projection and grad_out in bf16 next to fp32 reference points, the msda_*_fused_f32_sbf16 kernels)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msda_triton_amd import _lib, multiscale_deformable_attention, synth
from msda_triton_amd.functional import fused_module_core, module_sampling_inputs

dev = torch.device("cuda", 0)
args = [a for a in sys.argv[1:] if a != "--"]
storage = "storage" in args
vdt = torch.bfloat16 if ("value=bf16" in args or storage) else torch.float32
rd = 4 if "ref4" in args else 2
opts = [a for a in args if a == "-" or ("=" in a and not a.startswith("value="))] or ["-"]
wl = synth.WORKLOADS["c2_q10k"]
torch.manual_seed(0)
shapes = torch.tensor(wl.levels, device=dev)
value = torch.randn(wl.B, wl.I, wl.H, wl.D, device=dev).to(vdt).requires_grad_()
proj = torch.randn(wl.B, wl.Q, wl.H, wl.L, wl.P, 3, device=dev).to(vdt if storage else torch.float32).requires_grad_()
ref = torch.rand(wl.B, wl.Q, rd, device=dev)
go = torch.randn(wl.B, wl.Q, wl.H, wl.D, device=dev)
go_f = go.to(vdt) if storage else go
with torch.no_grad():
    pts, att = module_sampling_inputs(proj.float(), shapes, ref)
pts, att = pts.detach().requires_grad_(), att.detach().requires_grad_()


def fused():
    out = fused_module_core(value, shapes, proj, ref, wl.padding_mode, wl.align_corners)
    out.backward(go_f)
    value.grad = proj.grad = None


def plain():
    out = multiscale_deformable_attention(value, shapes, pts, att, wl.padding_mode, wl.align_corners)
    out.backward(go)
    value.grad = pts.grad = att.grad = None


def measure(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    _lib.set_option("profile", 1)
    _lib.profile_read()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    r = _lib.profile_read()
    _lib.set_option("profile", 0)
    return {k.replace("msda_", "").replace("_kernel", ""): round(v[1], 1) for k, v in r.items()}


keys = sorted({kv.split("=")[0] for o in opts if o != "-" for kv in o.split(",")})
defaults = {k: _lib.get_option(k) for k in keys}
for rep in range(2):
    for o in opts:
        for k, v in defaults.items():
            _lib.set_option(k, v)
        if o != "-":
            for kv in o.split(","):
                k, v = kv.split("=")
                _lib.set_option(k, int(v))
        print("fused %-14s" % o, measure(fused), flush=True)
    for k, v in defaults.items():
        _lib.set_option(k, v)
    print("plain %-14s" % "-", measure(plain), flush=True)
