"""dev tool: a pyramid with more bilinear cells than the count / place passes keep in LDS at once (several trips per
workgroup): per-group kernel times.  Usage: python tools/big_pyramid_bench.py [lib.so] [option=int ...]"""
import os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    shutil.copy(sys.argv[1], os.path.join(ROOT, "msda_triton_amd", "libmsda_hip.so"))
import torch
from msda_triton_amd import _lib
for kv in sys.argv[1:]:
    if "=" in kv:
        _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
from msda_triton_amd import synth
from msda_triton_amd.functional import KernelTimer, multiscale_deformable_attention

Q = int(os.environ.get("BIG_Q", "20000"))
B = int(os.environ.get("BIG_B", "1"))
LEVELS = {"256": ((256, 256), (128, 128), (64, 64), (32, 32)), "coco": ((100, 134), (50, 67), (25, 34), (13, 17))}[
    os.environ.get("BIG_PYR", "256")]
wl = synth.Workload("big", B, Q, 8, 32, LEVELS, 4, "float32", "zeros", False)
dev = torch.device("cuda", 0)
d = synth.make_inputs_torch(wl, dev, seed=0)
v, pts, att = d["value"].requires_grad_(True), d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)
go = torch.rand(wl.B, wl.Q, wl.H, wl.D, device=dev)
def step():
    multiscale_deformable_attention(v, d["shapes"], pts, att, wl.padding_mode, wl.align_corners).backward(go)
    v.grad = pts.grad = att.grad = None
for _ in range(5): step()
with KernelTimer() as kt:
    for _ in range(20): step()
torch.cuda.synchronize()
print({k: round(ms * 1e3, 1) for k, (n, ms) in kt.summary().items()})
