"""dev tool: fwd + bwd step at the c2 pyramid for (B, H) combinations whose B * H planes do or do not cover the eight XCDs
evenly, with the default block order, the linear one (xcd_map=0) and one plane per LDS-level workgroup (lds_planes=1) — what
plane_grid's rule (msda_launch.hpp) is based on.  (The first option of a line is measured on a colder GPU: compare the others.)
    python tools/few_planes_bench.py"""
import sys, os, time, dataclasses
sys.path.insert(0, os.getcwd())
import torch
from msda_triton_amd import _lib, synth, multiscale_deformable_attention
wl0 = synth.WORKLOADS["c2_q10k"]
for B, H, Q in ((1, 8, 40000), (1, 4, 40000), (3, 4, 10000), (1, 2, 40000), (2, 8, 10000), (5, 8, 5000)):
    wl = dataclasses.replace(wl0, B=B, H=H, Q=Q)
    d = synth.make_inputs_torch(wl, torch.device("cuda", 0), seed=0)
    img, shapes = d["value"].requires_grad_(True), d["shapes"]
    pts, attn = d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)
    def step():
        out = multiscale_deformable_attention(img, shapes, pts, attn, wl.padding_mode, wl.align_corners)
        out.backward(torch.rand_like(out)); img.grad = pts.grad = attn.grad = None
    res = {}
    for name, opts in (("default", {}), ("xcd_map=0", {"xcd_map": 0}), ("lds_planes=1", {"lds_planes": 1})):
        for k, v in opts.items(): _lib.set_option(k, v)
        for _ in range(30): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): step()
        torch.cuda.synchronize(); res[name] = round((time.perf_counter() - t0) / 30 * 1e3, 4)
        _lib.set_option("xcd_map", 1); _lib.set_option("lds_planes", 0)
    print("B=%d H=%d Q=%d (planes %d): step ms" % (B, H, Q, B * H), res, flush=True)
