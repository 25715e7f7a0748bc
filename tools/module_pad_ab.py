"""dev tool (GPU box): the nn.Module's training step with the value projection written DENSE (MSDA_VALUE_ROW_PAD=0) and
with padded rows (the default rule, functional.value_row_pad), alternated inside one process: step ms and the device
times of the library's kernels (its own event pairs).
    python tools/module_pad_ab.py [--queries 10000 2500 900] [--fp32] [--rounds 3]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msda_triton_amd import MultiscaleDeformableAttention, _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, nargs="+", default=[10000, 2500, 900])
ap.add_argument("--fp32", action="store_true")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=40)
args = ap.parse_args()
dev = torch.device("cuda", 0)
wl = synth.WORKLOADS["c2_q10k"]
shapes = torch.tensor(wl.levels, device=dev)
EMB = wl.H * wl.D
torch.manual_seed(0)
m = MultiscaleDeformableAttention(EMB, EMB, wl.L, wl.H, wl.P, wl.padding_mode, wl.align_corners).to(dev)
img = torch.randn(wl.B, wl.I, EMB, device=dev, requires_grad=True)
for Q in args.queries:
    q = torch.randn(wl.B, Q, EMB, device=dev, requires_grad=True)
    ref = torch.rand(wl.B, Q, 2, device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.fp32):
            out = m(img, shapes, q, ref)
        out.float().sum().backward()
        m.zero_grad(set_to_none=True)
        img.grad = q.grad = None

    def measure():
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.steps
        _lib.set_option("profile", 1)
        _lib.profile_read()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        k = {n.replace("msda_", "").replace("_kernel", ""): round(v[1], 1) for n, v in _lib.profile_read().items()}
        _lib.set_option("profile", 0)
        return round(ms, 4), k

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        step()
    for _ in range(args.rounds):
        for label, env in (("dense", "0"), ("padded", None)):
            if env is None:
                os.environ.pop("MSDA_VALUE_ROW_PAD", None)
            else:
                os.environ["MSDA_VALUE_ROW_PAD"] = env
            ms, k = measure()
            print(f"Q={Q:6d} {'fp32' if args.fp32 else 'bf16'} {label:7s} step {ms} ms", json.dumps(k), flush=True)
    os.environ.pop("MSDA_VALUE_ROW_PAD", None)
