"""dev tool (GPU box): where the row-sharded operator's world-1 step differs from the direct operator's.

One process, RCCL initialised in-process at world size 1; the routes take turns (same box, same clocks):
  direct      multiscale_deformable_attention (C++ autograd node when the binding is built)
  pyfn        the Python autograd Function + ctypes launchers
  emu         row_sharded_…(compute_only_as=(1, 0))            (no process group involved)
  nccl1       row_sharded_… over the nccl group, one piece
  nccl4       … four pieces (no exchange at one rank: the pieces' launches only; the C++ row node since round 6)
  loop1/loop4 the Python node with its collectives run against the rank itself (loopback=True)
For every route: wall ms per step (K steps between synchronisations), host ms per step (time to ENQUEUE the K steps)
and the library's own per-kernel device times (msda_profile_read).

  python tools/shard_overhead.py [--workload c2_q10k] [--steps 50] [--rounds 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2_q10k")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--routes", default="direct,pyfn,emu,nccl1,nccl4,loop1,loop4")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from msda_triton_amd import _lib, synth
    from msda_triton_amd.distributed import row_sharded_multiscale_deformable_attention
    from msda_triton_amd.functional import (_HipMultiscaleDeformableAttentionFunction,
                                            multiscale_deformable_attention)

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    wl = synth.WORKLOADS[args.workload]
    pm, ac = wl.padding_mode, wl.align_corners
    rows = wl.B * wl.Q
    d = synth.make_inputs_torch(wl, dev, seed=0, rows=(0, rows))
    img, shapes = d["value"].requires_grad_(True), d["shapes"]
    pts, att = d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)
    pts4, att4 = pts.view(wl.B, wl.Q, *pts.shape[1:]), att.view(wl.B, wl.Q, *att.shape[1:])

    def sharded(**kw):
        return row_sharded_multiscale_deformable_attention(img, shapes, pts, att, pm, ac, inputs_are_sharded=True,
                                                           num_queries=wl.Q, grad_value_sync="owners", **kw)

    ops = {
        "direct": lambda: multiscale_deformable_attention(img, shapes, pts4, att4, pm, ac),
        "pyfn": lambda: _HipMultiscaleDeformableAttentionFunction.apply(img, shapes, pts4, att4, pm, ac, 0),
        "emu": lambda: sharded(compute_only_as=(1, 0), overlap_chunks=1),
        "nccl1": lambda: sharded(overlap_chunks=1),
        "nccl4": lambda: sharded(overlap_chunks=4),
        "loop1": lambda: sharded(overlap_chunks=1, loopback=True),   # the Python node + in-place all-gather on RCCL
        "loop4": lambda: sharded(overlap_chunks=4, loopback=True),   # ... + four rounds of self send / receive
    }
    routes = [r for r in args.routes.split(",") if r in ops]

    def step(op):
        out = op()
        out.backward(torch.rand_like(out))
        img.grad = pts.grad = att.grad = None

    def timed(op, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(op)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t2 - t0) * 1e3 / n, (t1 - t0) * 1e3 / n

    # spin-up: ~200 ms of steps
    for r in routes:
        for _ in range(5):
            step(ops[r])
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        step(ops[routes[0]])
    res = {r: {"wall_ms": [], "host_ms": []} for r in routes}
    for _ in range(args.rounds):
        for r in routes:
            for _ in range(10):
                step(ops[r])
            w, h = timed(ops[r], args.steps)
            res[r]["wall_ms"].append(round(w, 4))
            res[r]["host_ms"].append(round(h, 4))
    # per-kernel device times
    _lib.set_option("profile", 1)
    for r in routes:
        for _ in range(5):
            step(ops[r])
        torch.cuda.synchronize()
        _lib.profile_read()
        for _ in range(20):
            step(ops[r])
        torch.cuda.synchronize()
        res[r]["kernels_us"] = {k: round(v[1], 2) for k, v in _lib.profile_read().items()}
        res[r]["kernels_sum_us"] = round(sum(res[r]["kernels_us"].values()), 1)
    _lib.set_option("profile", 0)
    for r in routes:
        print(r, json.dumps(res[r]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
