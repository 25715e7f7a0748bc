#!/usr/bin/env python3
"""Dev tool: time fwd / bwd kernels on every BASELINE workload (random device-side inputs)."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msda_triton_amd import synth, _lib
from msda_triton_amd.functional import KernelTimer, msda_hip_fwd, msda_hip_bwd

dev = "cuda:0"
opts = [a for a in sys.argv[1:] if "=" in a and ":" not in a]
for kv in opts:
    k, v = kv.split("=")
    _lib.set_option(k, int(v))
names = [a for a in sys.argv[1:] if "=" not in a or ":" in a] or list(synth.WORKLOADS)
for name in names:
    if ":" in name:  # e.g. c2_q10k:B=1:Q=40000 — a BASELINE workload with fields overridden
        import dataclasses
        base, *over = name.split(":")
        wl = dataclasses.replace(synth.WORKLOADS[base], **{k: int(v) for k, v in (o.split("=") for o in over)})
    else:
        wl = synth.WORKLOADS[name]
    dt = getattr(torch, os.environ.get("MSDA_SWEEP_DTYPE", wl.dtype))
    torch.manual_seed(0)
    v = torch.randn(wl.B, wl.I, wl.H, wl.D, device=dev, dtype=torch.float32).to(dt)
    s = torch.tensor(wl.levels, device=dev)
    lo, hi = (float(x) for x in os.environ.get("MSDA_SWEEP_LOC", "0,1").split(","))  # clustered sampling: e.g. 0.45,0.55
    l = (lo + (hi - lo) * torch.rand(wl.B, wl.Q, wl.H, wl.L, wl.P, 2, device=dev, dtype=torch.float32)).to(dt)
    a = torch.softmax(torch.randn(wl.B, wl.Q, wl.H, wl.L, wl.P, device=dev), -1).to(dt)
    g = torch.rand(wl.B, wl.Q, wl.H, wl.D, device=dev, dtype=torch.float32).to(dt)
    pm, ac = wl.padding_mode, wl.align_corners
    for _ in range(3):
        msda_hip_fwd(v, s, l, a, pm, ac)
        msda_hip_bwd(g, v, s, l, a, pm, ac)
    torch.cuda.synchronize()
    with KernelTimer() as kt:
        for _ in range(10):
            msda_hip_fwd(v, s, l, a, pm, ac)
            msda_hip_bwd(g, v, s, l, a, pm, ac)
        torch.cuda.synchronize()
    res = {k: round(ms * 1e3, 1) for k, (n, ms) in kt.summary().items()}
    fwd_us = res["msda_fwd"]
    print(json.dumps({"workload": name, "dtype": wl.dtype, "us": res,
                      "fwd_alg_GBs": round(wl.alg_fwd_bytes / fwd_us / 1e3, 1),
                      "fwd_gather_TBs": round(wl.gather_fwd_bytes / fwd_us / 1e6, 2),
                      "ws_MB": round(_lib.load().msda_bwd_workspace_bytes(wl.B, wl.I, wl.H, wl.D, wl.Q, wl.L, wl.P, v.element_size()) / 1e6, 1)}))
    del v, l, a, g
    torch.cuda.empty_cache()
