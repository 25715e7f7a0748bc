"""dev tool: decoder-sized calls over a real-image pyramid with and without the level-size hint (level_shapes=)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msda_triton_amd import synth, multiscale_deformable_attention
from msda_triton_amd.functional import KernelTimer
dev = torch.device("cuda")
rows = {}
for name, B, Q in (("dec_coco B=8 Q=900", 8, 900), ("dec_coco B=2 Q=900", 2, 900), ("dec_coco B=8 Q=300", 8, 300), ("dec_coco B=16 Q=900", 16, 900)):
    wl0 = synth.WORKLOADS["dec_coco"]
    wl = synth.Workload("x", B, Q, wl0.H, wl0.D, wl0.levels, wl0.P, "float32", "zeros", False)
    d = synth.make_inputs_torch(wl, dev, seed=0)
    v, l, a = (d[k].requires_grad_(True) for k in ("value", "loc", "attn"))
    g = d["grad_out"]
    res = {}
    for label, ls in (("without", None), ("with level_shapes", list(wl.levels))):
        def step():
            multiscale_deformable_attention(v, d["shapes"], l, a, "zeros", False, level_shapes=ls).backward(g)
            v.grad = l.grad = a.grad = None
        for _ in range(10): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): step()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 200 * 1e3
        with KernelTimer() as kt:
            for _ in range(20): step()
            torch.cuda.synchronize()
        res[label] = {"fwd_bwd_ms": round(ms, 4), "kernels_us": {k: round(msk * 1e3, 1) for k, (n, msk) in kt.summary().items()}}
    rows[name] = res
    print(name, json.dumps(res), flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
