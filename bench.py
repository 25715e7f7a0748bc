#!/usr/bin/env python3
"""bench.py — MSDA fwd and fwd+bwd at 10k queries on MI355X (BASELINE.json `metric`).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One *step* = the body of the reference benchmark (scripts/benchmark.py:90-94 of rziga/msda-triton):
    out = op(img, shapes, pts, attn, padding_mode, align_corners)
    out.backward(torch.rand_like(out)); grads = None
through the public autograd API, on BASELINE configs[1] at Q = 10 000 (B=4, H=8, D=32, L=4 levels
64..8, P=4, fp32, border / align_corners=True), synthetic inputs resident in HBM.

N > 1 (weak scaling): every rank owns B*Q rows of a N times larger problem (global Q = N*10 000) against the
replicated value pyramid; the kernels write the local rows straight into the result, the peers' rows arrive by an
in-place exchange that overlaps the next piece's kernels (RCCL); grad_value is summed only among the ranks that
share a batch element (none while N divides B).  `value` = query rows (b, q) processed per second by the whole
job, fwd+bwd.  The same run also times a STRONG-scaling leg under the key `strong_scaling_c5`: BASELINE configs[4]
(B=4, Q=100 000, D=64, L=5, P=8, fp16) with its 400 000 rows split over the N ranks.

The N=1 run also carries (same JSON line):
  * `do_bench`: the reference's timing recipe for this workload (scripts/benchmark.py:38,52-54,90-94 — triton.testing.do_bench:
    >= 100 ms of warm-up, >= 1 s of repetitions, per-repetition device events, median + p20 / p80), once with the
    caches flushed between repetitions (cold, as do_bench does) and once back to back (warm);
  * `configs`: every other BASELINE.json config that fits one GPU (c1, c2 @ 1k / 5k, c3, c4): fwd_ms, fwd_bwd_ms,
    per-launch-group microseconds with their roofline fraction and measured HBM traffic, peak memory.
`ms_per_step` stays the K timed steps of the headline workload.

`--backend gloo --device cpu` (with `--workload dryrun`) runs the whole N-rank control flow — sharding, both exchanges,
the stall guard, the strong-scaling leg's plumbing, the JSON contract — on host tensors over gloo: a rehearsal of the
multi-GPU path on a box without GPUs (tests/test_distributed_cpu.py runs it at world size 2).  Its numbers mean nothing.

Rank 0 prints ONE JSON line.  An optional leg that fails is reported under its key as {"error": ...} and the process
exits with status 1 AFTER the line has been printed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# This pool's host driver only supports dmabuf IPC: without it RCCL's buffer exchange between the ranks' processes fails with
# hipIpcGetMemHandle: invalid argument.  The boxes export it already; set before anything initialises the runtime.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
README_RTX2060_MS = {"fwd": 3.78, "fwd_bwd": 22.78}  # reference README.md:18-19 (Triton, RTX 2060)
L1_PEAK_GBS = 64 * 256 * 2.4  # vector L1: 64 B/clk/CU x 256 CUs x 2.4 GHz = 39 322 GB/s (guides/MI355X_MICROARCH.md)
LDS_PEAK_GBS = 256 * 256 * 2.4  # LDS, ds_read_b128: 256 B/clk/CU x 256 CUs x 2.4 GHz = 157 286 GB/s (same guide, "LDS")


def lds_row_fraction(wl, info):
    """Fraction of the forward's gathered rows that come from LDS: the kernel variant with LDS-served levels
    (msda_last_launch_info: fwd_variant 1) keeps the longest SUFFIX of the level list whose pixels fit
    `fwd_lds_level_bytes` (msda_kernels.hpp coarse_levels — restated here); every level has P samples per unit."""
    if not info or info.get("fwd_variant") != 1:
        return 0.0
    cap = info["fwd_lds_level_bytes"] // (wl.D * wl.elem_size)
    pixels, first = 0, wl.L
    for lvl in range(wl.L - 1, -1, -1):
        n = wl.levels[lvl][0] * wl.levels[lvl][1]
        if n > cap - pixels:
            break
        pixels += n
        first = lvl
    return (wl.L - first) / wl.L


def single_kernel_alg_bytes(wl):
    """Compulsory bytes per launch of the individual kernels that have a clean definition (the forward and the
    sample-gradient kernel are their launch groups; the grad_value pipeline's kernels share one budget, ALG of the
    group, and are listed with their times only)."""
    alg = kernel_alg_bytes(wl)
    return {"msda_fwd_kernel": alg["msda_fwd"], "msda_bwd_sample_kernel": alg["msda_bwd_sample"]}



def kernel_alg_bytes(wl):
    """Compulsory bytes per launch of each kernel (every operand touched once)."""
    s = wl.elem_size
    BIHD = wl.B * wl.I * wl.H * wl.D
    S = wl.B * wl.Q * wl.H * wl.L * wl.P
    BQHD = wl.B * wl.Q * wl.H * wl.D
    return {
        "msda_fwd": s * (BIHD + 3 * S + BQHD) + 16 * wl.L,           # == BASELINE.md ALG_FWD
        "msda_bwd_sample": s * (BIHD + 3 * S + BQHD + 3 * S),         # reads value, loc, attn, grad_out; writes grad_loc, grad_attn
        "msda_bwd_value": s * (3 * S + BQHD + BIHD),                  # reads loc, attn, grad_out; writes grad_value
    }


def cpu_baseline(wl, budget_s=20.0):
    """CPU baselines on the GPU box's host cores, same workload.  Top level: the package's own
    PyTorch-native host path (`functional.native_multiscale_deformable_attention`, the role of the reference's
    frontend.py:15-68 fallback) through autograd, kind "native".  Beside it (`c_port`): the CPU oracle — a C port of
    the reference algorithm, OpenMP over the host cores — whole fwd+bwd passes until ~budget_s of CPU work is spent."""
    import numpy as np
    from msda_triton_amd import synth
    from oracle import msda_oracle

    q_sample = min(wl.Q, 10000)
    d = synth.make_inputs_numpy(wl, seed=0, q_end=q_sample)
    a = {k: (v if k == "shapes" else np.ascontiguousarray(v, dtype=np.float32)) for k, v in d.items()}
    pm, ac = wl.padding_mode, wl.align_corners
    msda_oracle.forward(a["value"], a["shapes"], a["loc"], a["attn"], pm, ac)  # warm-up (page-in, omp pool)
    t_fwd, t_all, n = [], [], 0
    t_start = time.perf_counter()
    while n < 3 or (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        msda_oracle.forward(a["value"], a["shapes"], a["loc"], a["attn"], pm, ac)
        t1 = time.perf_counter()
        msda_oracle.backward(a["grad_out"], a["value"], a["shapes"], a["loc"], a["attn"], pm, ac)
        t2 = time.perf_counter()
        t_fwd.append(t1 - t0)
        t_all.append(t2 - t0)
        n += 1
        if n >= 30:
            break
    med_all = sorted(t_all)[len(t_all) // 2]
    med_fwd = sorted(t_fwd)[len(t_fwd) // 2]
    port = {
        "value": wl.B * q_sample / med_all,
        "unit": "queries/s",
        "cores": msda_oracle.num_threads(),
        "kind": "port",
        "sample": f"{n} fwd+bwd passes of {wl.name} restricted to the first {q_sample} of {wl.Q} queries per batch "
                  f"element (B={wl.B}), fp32, median",
        "fwd_ms_scaled_to_full": med_fwd * 1e3 * wl.Q / q_sample,
        "fwd_bwd_ms_scaled_to_full": med_all * 1e3 * wl.Q / q_sample,
    }
    try:
        native = torch_cpu_fallback(wl)
    except Exception as e:  # the extra figure must never cost the bench its JSON line: the C port then stands alone
        port["native_error"] = repr(e)[:300]
        return port
    return {
        "value": wl.B * native["q_sample"] / (native["fwd_bwd_ms"] * 1e-3),
        "unit": "queries/s",
        "cores": native["threads"],
        "kind": "native",
        "what": "msda_triton_amd.functional.native_multiscale_deformable_attention (plain PyTorch on host tensors, the "
                "reference fallback's role, frontend.py:15-68) through autograd",
        "sample": native["sample"],
        "fwd_bwd_ms_scaled_to_full": native["fwd_bwd_ms_scaled_to_full"],
        "c_port": port,
    }


def torch_cpu_fallback(wl, budget_s=8.0, q_sample=10000):
    """The host-tensor path of this package (plain PyTorch, the role of the reference's CPU fallback,
    frontend.py:15-68) timed on the host cores through autograd — at the workload's full query count up to 10 000
    (no extrapolation: the value-plane cost does not scale with the queries)."""
    import torch
    from msda_triton_amd import synth
    from msda_triton_amd.functional import native_multiscale_deformable_attention

    q_sample = min(wl.Q, q_sample)
    d = synth.make_inputs_torch(wl, "cpu", seed=0, dtype=torch.float32, q_end=q_sample)
    v, l, a = (d[k].requires_grad_(True) for k in ("value", "loc", "attn"))
    times, t_start = [], time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        out = native_multiscale_deformable_attention(v, d["shapes"], l, a, wl.padding_mode, wl.align_corners)
        out.backward(d["grad_out"])
        v.grad = l.grad = a.grad = None
        times.append(time.perf_counter() - t0)
        if len(times) >= 10:
            break
    med = sorted(times)[len(times) // 2]
    return {"fwd_bwd_ms": med * 1e3, "q_sample": q_sample,
            "fwd_bwd_ms_scaled_to_full": med * 1e3 * wl.Q / q_sample, "threads": torch.get_num_threads(),
            "sample": f"{len(times)} fwd+bwd passes of {wl.name} on the first {q_sample} of {wl.Q} queries per batch "
                      f"element (B={wl.B}), fp32, median"}


def _quantile(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    k = (len(xs) - 1) * q
    lo, hi = int(k), min(int(k) + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (k - lo)


def do_bench(fn, dev, flush, warmup_ms=100.0, rep_ms=1000.0):
    """triton.testing.do_bench semantics with HIP events (reference scripts/benchmark.py:38,52-54): estimate the
    cost of one call, run ~warmup_ms of warm-up, then ~rep_ms worth of repetitions, each bracketed by its own event
    pair; `flush`: a buffer larger than L2 + Infinity Cache is rewritten before every repetition (do_bench's cache
    clear), so every repetition starts cold.  Returns median / p20 / p80 in ms."""
    import torch

    cache = torch.empty(512 * 1024 * 1024, dtype=torch.int8, device=dev) if flush else None
    fn()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        if cache is not None:
            cache.zero_()
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    est = e0.elapsed_time(e1) / 5
    n_warm = max(1, int(warmup_ms / est))
    n_rep = max(10, min(5000, int(rep_ms / est)))
    for _ in range(n_warm):
        fn()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(n_rep)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(n_rep)]
    for i in range(n_rep):
        if cache is not None:
            cache.zero_()
        starts[i].record()
        fn()
        ends[i].record()
    torch.cuda.synchronize(dev)
    t = [a.elapsed_time(b) for a, b in zip(starts, ends)]
    return {"median_ms": _quantile(t, 0.5), "p20_ms": _quantile(t, 0.2), "p80_ms": _quantile(t, 0.8), "reps": n_rep,
            "warmup_calls": n_warm}


def load_traffic(workload):
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            return json.load(f).get(workload, {})
    return {}


def kernel_table(wl, kern, traffic):
    """per launch group: launches, mean us (HIP events), algorithmic bytes, achieved GB/s, fraction of the HBM peak,
    HBM traffic per launch measured by the PMC passes (profiles/hbm_traffic.json; None if not profiled)"""
    alg = kernel_alg_bytes(wl)
    out = {}
    for name, (n, mean_ms) in sorted(kern.items()):
        gbs = alg[name] / (mean_ms * 1e-3) / 1e9
        out[name] = {"launches": n, "avg_us": round(mean_ms * 1e3, 2), "alg_bytes": alg[name],
                     "achieved_GBs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                     "traffic": traffic.get(name)}
    return out


def bench_config(name, dev, steps=20, warmup=5):
    """One BASELINE config on one GPU through the public autograd API: warm mean of `steps` steps (fwd only, fwd+bwd),
    per-launch-group HIP-event times, peak memory of a fwd+bwd."""
    import torch
    from msda_triton_amd import synth
    from msda_triton_amd.functional import KernelTimer, multiscale_deformable_attention

    wl = synth.WORKLOADS[name]
    d = synth.make_inputs_torch(wl, dev, seed=0)
    img, shapes = d["value"].requires_grad_(True), d["shapes"]
    pts, attn = d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)
    pm, ac = wl.padding_mode, wl.align_corners

    def step():
        out = multiscale_deformable_attention(img, shapes, pts, attn, pm, ac)
        out.backward(torch.rand_like(out))
        img.grad = pts.grad = attn.grad = None

    def fwd_only():
        with torch.no_grad():
            multiscale_deformable_attention(img, shapes, pts, attn, pm, ac)

    def timed_once(fn, n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) * 1e3 / n

    def timed(fn, n):
        # median of three means of n steps: one stall of the box (an allocation, another tenant) inside a 5 ms window
        # otherwise lands in the figure — seen once as 1.79 ms for a 0.25 ms step
        return sorted(timed_once(fn, n) for _ in range(3))[1]

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    base = torch.cuda.memory_allocated(dev)
    step()
    torch.cuda.synchronize(dev)
    peak = torch.cuda.max_memory_allocated(dev)
    ms_step = timed(step, steps)
    for _ in range(warmup):
        fwd_only()
    ms_fwd = timed(fwd_only, steps)
    # the same kernels through the launcher API (msda_hip_fwd / msda_hip_bwd: no autograd engine, grad_out given):
    # what the step costs a caller that drives the C ABI itself; the difference to fwd_bwd_ms is PyTorch's engine
    # start-up, three AccumulateGrad nodes and rand_like
    from msda_triton_amd.functional import msda_hip_bwd, msda_hip_fwd
    go = torch.rand_like(d["grad_out"])
    iv, ip, ia = img.detach(), pts.detach(), attn.detach()

    def launchers():
        msda_hip_fwd(iv, shapes, ip, ia, pm, ac)
        msda_hip_bwd(go, iv, shapes, ip, ia, pm, ac)

    for _ in range(warmup):
        launchers()
    ms_launchers = timed(launchers, steps)
    with KernelTimer() as kt:
        timed_once(step, steps)
    kernels = kernel_table(wl, kt.summary(), load_traffic(name))
    return {"workload": f"{wl.name}: B={wl.B} Q={wl.Q} H={wl.H} D={wl.D} L={wl.L} levels={list(wl.levels)} P={wl.P} "
                        f"{wl.dtype} {pm} align_corners={ac}",
            "steps": steps, "timing": "median of three warm means of `steps` steps",
            "fwd_ms": ms_fwd, "fwd_bwd_ms": ms_step, "launcher_api_fwd_bwd_ms": ms_launchers,
            "alg_fwd_bwd_GBs": round((wl.alg_fwd_bytes + wl.alg_bwd_bytes) / (ms_step * 1e-3) / 1e9, 1),
            "peak_mem_MB": round(peak / 1e6, 1), "inputs_MB": round(base / 1e6, 1), "kernels": kernels}


def padded_rows_leg(wl_name, dev, steps=50, warmup=10, rounds=3):
    """NOT the headline: the same step with `img` handed over in PADDED rows (every pixel's H * D channels one 128-byte line
    apart: functional.padded_value_rows, C ABI value_row_stride) next to the dense layout the reference's callers produce,
    alternated in this process.  What a caller that owns the layout of its value projection gets."""
    import torch
    from msda_triton_amd import _lib, synth
    from msda_triton_amd.functional import multiscale_deformable_attention, padded_value_rows

    wl = synth.WORKLOADS[wl_name]
    d = synth.make_inputs_torch(wl, dev, seed=0)
    dense = d["value"]
    padded = padded_value_rows(*dense.shape, dense.dtype, dense.device)
    padded.copy_(dense)
    shapes, pts, attn = d["shapes"], d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)
    pm, ac = wl.padding_mode, wl.align_corners
    out = {"what": "the headline step with `img` in padded rows (a caller that owns the layout) against the dense layout; "
                   "NOT the headline — the reference's callers hand over dense tensors",
           "value_row_stride": padded.stride(1) * padded.element_size()}
    res = {"dense": [], "padded": [], "padded_copy_per_step": []}
    kern = {}
    for _ in range(rounds):
        # padded_copy_per_step: the ONE-COPY variant for callers with a dense tensor — the step first copies `img` into
        # padded rows (what a forward that made the copy itself would pay: 22 MB read, 25 MB written at the c2 shape)
        for name, img in (("dense", dense), ("padded", padded), ("padded_copy_per_step", padded)):
            img = img.detach().requires_grad_(True)
            copy = name == "padded_copy_per_step"

            def step():
                if copy:
                    with torch.no_grad():
                        img.copy_(dense)
                o = multiscale_deformable_attention(img, shapes, pts, attn, pm, ac)
                o.backward(torch.rand_like(o))
                img.grad = pts.grad = attn.grad = None

            for _ in range(warmup):
                step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize(dev)
            res[name].append(round((time.perf_counter() - t0) * 1e3 / steps, 4))
            _lib.set_option("profile", 1)
            try:
                _lib.profile_read()
                for _ in range(10):
                    step()
                torch.cuda.synchronize(dev)
                kern[name] = {k: round(v[1], 1) for k, v in _lib.profile_read().items() if "fwd" in k or "bwd_sample" in k}
            finally:
                _lib.set_option("profile", 0)
    out["fwd_bwd_ms"] = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    out["fwd_bwd_ms_rounds"] = res
    out["kernel_us"] = kern
    return out


def real_pyramid_leg(dev, steps=50, warmup=10):
    """Not a BASELINE config: the c4 decoder call (B=8, Q=900) over the pyramid of an 800 x 1066 image (c3's levels)
    instead of 64 x 64 ... 8 x 8 — what a Grounding-DINO / Deformable-DETR decoder layer sees at COCO size — without
    and with the level sizes given as host numbers (`level_shapes=`, include/msda_hip.h max_level_cells)."""
    import torch
    from msda_triton_amd import synth
    from msda_triton_amd.functional import multiscale_deformable_attention

    wl = synth.WORKLOADS["dec_coco"]
    d = synth.make_inputs_torch(wl, dev, seed=0)
    v, l, a = (d[k].requires_grad_(True) for k in ("value", "loc", "attn"))
    out = {"workload": f"dec_coco: B={wl.B} Q={wl.Q} H={wl.H} D={wl.D} levels={list(wl.levels)} P={wl.P} {wl.dtype} "
                       f"{wl.padding_mode} align_corners={wl.align_corners} (not a BASELINE config)"}
    for key, ls in (("fwd_bwd_ms", None), ("fwd_bwd_ms_with_level_shapes", list(wl.levels))):
        def step():
            o = multiscale_deformable_attention(v, d["shapes"], l, a, wl.padding_mode, wl.align_corners, level_shapes=ls)
            o.backward(torch.rand_like(o))
            v.grad = l.grad = a.grad = None
        for _ in range(warmup):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        out[key] = (time.perf_counter() - t0) * 1e3 / steps
    return out


def triton_comparator_leg(wl_name, dev):
    """A Triton kernel WRITTEN FOR THIS REPO (scripts/triton_comparator.py: the reference's parallelisation — a program
    per (query, batch, head), tl.atomic_add grad_value, num_warps autotuned — but not its code, which cannot travel to
    the GPU box) timed next to the HIP operator on the same GPU, SURVEY.md 8d.  A comparator, not a product path: any
    failure (no Triton, compile error) is reported here and costs the bench nothing."""
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location(
            "msda_triton_comparator", os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "triton_comparator.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if not mod.HAVE_TRITON:
            return {"unavailable": "Triton is not importable"}
        import triton
        r = mod.compare(wl_name, dev)
        return {"what": "builder-authored Triton comparator (scripts/triton_comparator.py), NOT the reference's kernel; "
                        "same inputs, same GPU, warm mean of 20 steps each",
                "triton_version": triton.__version__,
                "triton_fwd_ms": r["triton"]["fwd_ms"], "triton_fwd_bwd_ms": r["triton"]["fwd_bwd_ms"],
                "triton_fwd_bwd_ms_relaxed_atomics": r["triton_relaxed_atomics"]["fwd_bwd_ms"],
                "hip_fwd_ms": r["hip"]["fwd_ms"], "hip_fwd_bwd_ms": r["hip"]["fwd_bwd_ms"],
                "hip_speedup": {"fwd": r["hip_speedup"]["fwd_ms"], "fwd_bwd": r["hip_speedup"]["fwd_bwd_ms"],
                                "fwd_bwd_vs_relaxed_atomics": r["hip_speedup"]["fwd_bwd_ms_vs_relaxed_atomics"]},
                "atomics": "tl.atomic_add with its default acq_rel ordering, as the reference calls it "
                           "(kernels.py:550-553); the relaxed figure is the same kernel with sem='relaxed'",
                "triton_num_warps": r["triton_num_warps"], "max_abs_diff_vs_hip": r["max_abs_diff"]}
    except Exception as e:  # noqa: BLE001
        return {"unavailable": repr(e)[:300]}


def strong_scaling_leg(wl_name, dev, world, rank, use_dist, chunks=None, steps=5, warmup=2):
    """A BASELINE config (configs[4], the stress shape, in real runs) with its B*Q rows split over the ranks: fwd+bwd
    ms per step.  Inputs are drawn on the device (torch RNG, same seed on every rank for the replicated value pyramid):
    only shapes matter."""
    import torch
    import torch.distributed as dist
    from msda_triton_amd import synth
    from msda_triton_amd.distributed import row_shard_bounds, row_sharded_multiscale_deformable_attention
    from msda_triton_amd.functional import multiscale_deformable_attention

    wl = synth.WORKLOADS[wl_name]
    on_gpu = torch.device(dev).type == "cuda"
    dt = getattr(torch, wl.dtype) if on_gpu else torch.float32  # (the host path is a dry run: fp32)
    rows = wl.B * wl.Q
    r0, r1 = row_shard_bounds(rows, world, rank) if use_dist else (0, rows)
    g = torch.Generator(device=dev).manual_seed(1234)
    value = torch.randn(wl.B, wl.I, wl.H, wl.D, device=dev, generator=g).to(dt).requires_grad_(True)
    g.manual_seed(99 + rank)
    n = r1 - r0
    pts = torch.rand(n, wl.H, wl.L, wl.P, 2, device=dev, generator=g).to(dt).requires_grad_(True)
    att = torch.softmax(torch.randn(n, wl.H, wl.L * wl.P, device=dev, generator=g), -1).reshape(n, wl.H, wl.L, wl.P).to(dt)
    att.requires_grad_(True)
    shapes = torch.tensor(wl.levels, device=dev)

    def step():
        if use_dist:
            out = row_sharded_multiscale_deformable_attention(value, shapes, pts, att, wl.padding_mode, wl.align_corners,
                                                              inputs_are_sharded=True, num_queries=wl.Q,
                                                              grad_value_sync="owners", overlap_chunks=chunks)
        else:
            out = multiscale_deformable_attention(value, shapes, pts.view(wl.B, wl.Q, *pts.shape[1:]),
                                                  att.view(wl.B, wl.Q, *att.shape[1:]), wl.padding_mode, wl.align_corners)
        out.backward(torch.rand_like(out))
        value.grad = pts.grad = att.grad = None

    def barrier():
        if on_gpu:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt_s = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt_s], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_s = float(t.item())
    ms = dt_s * 1e3 / steps
    out = {"workload": f"{wl.name}: B={wl.B} Q={wl.Q} (global) H={wl.H} D={wl.D} L={wl.L} P={wl.P} {wl.dtype}, "
                       f"{rows} rows over {world} rank(s)", "scaling": "strong", "n_gpus": world, "steps": steps,
           "warmup": warmup, "ms_per_step": ms, "value": rows / (ms * 1e-3), "unit": "queries/s",
           "alg_fwd_bwd_GBs": round((wl.alg_fwd_bytes + wl.alg_bwd_bytes) / (ms * 1e-3) / 1e9, 1)}
    # speed-up against the one-GPU time of the same leg: this run's own at N = 1; at N > 1 the committed N = 1 record
    # (another run on another box: labelled) AND the same leg run unsharded on rank 0 of THIS job while the peers wait
    ref = os.path.join(ROOT, "profiles", "n1_reference.json")
    if world == 1:
        out["speedup_vs_n1"] = 1.0
        return out
    if os.path.exists(ref):
        with open(ref) as f:
            n1 = json.load(f).get(wl_name, {}).get("strong_leg_ms_per_step")
        if n1:
            out["n1_ms_per_step"] = n1
            out["speedup_vs_n1"] = n1 / ms
            out["n1_source"] = "profiles/n1_reference.json (an earlier --gpus 1 run of this bench on one MI355X: cross-run)"
    same_job = None
    if rank == 0:
        try:
            del pts, att
            g.manual_seed(98)
            pts = torch.rand(rows, wl.H, wl.L, wl.P, 2, device=dev, generator=g).to(dt).requires_grad_(True)
            att = torch.softmax(torch.randn(rows, wl.H, wl.L * wl.P, device=dev, generator=g), -1)
            att = att.reshape(rows, wl.H, wl.L, wl.P).to(dt).requires_grad_(True)

            def whole():
                o = multiscale_deformable_attention(value, shapes, pts.view(wl.B, wl.Q, *pts.shape[1:]),
                                                    att.view(wl.B, wl.Q, *att.shape[1:]), wl.padding_mode, wl.align_corners)
                o.backward(torch.rand_like(o))
                value.grad = pts.grad = att.grad = None

            whole()
            if on_gpu:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                whole()
            if on_gpu:
                torch.cuda.synchronize()
            same_job = (time.perf_counter() - t0) * 1e3 / steps
        except Exception as e:  # noqa: BLE001  (the peers wait at the barrier below whatever happens here)
            out["n1_same_job_error"] = repr(e)[:200]
    barrier()
    if same_job:
        out["n1_same_job_ms"] = same_job
        out["speedup_vs_n1_same_job"] = same_job / ms
    return out


def shard_compute_leg(dev, strong_name, weak_name, ns=(2, 4, 8), steps=5, warmup=2):
    """The COMPUTE half of the multi-GPU scaling model, measured on one GPU: for N in `ns`, rank 0's
    rows of the strong-scaling problem (the stress shape: its B*Q rows split N ways) and of the weak-scaling problem
    (the headline shape: Q x N queries per batch element, B*Q rows per rank) run through the row-sharded operator with
    every exchange left out (``compute_only_as=(N, 0)``): forward / forward+backward ms and ``t(1) / t_shard(N)`` —
    the speed-up ceiling before any byte crosses xGMI (strong: ideal N; weak: ideal 1) — next to the bytes of grad_value
    the rank would take into a sum with its peers."""
    import dataclasses

    import torch
    from msda_triton_amd import synth
    from msda_triton_amd.distributed import (default_overlap_chunks, owners_sum_bytes, row_shard_bounds,
                                             row_sharded_multiscale_deformable_attention)

    on_gpu = torch.device(dev).type == "cuda"

    def time_shard(wl, world, rank=0, chunks=None):
        dt = getattr(torch, wl.dtype) if on_gpu else torch.float32
        rows = wl.B * wl.Q
        r0, r1 = row_shard_bounds(rows, world, rank)
        n = r1 - r0
        g = torch.Generator(device=dev).manual_seed(4321)
        value = torch.randn(wl.B, wl.I, wl.H, wl.D, device=dev, generator=g).to(dt).requires_grad_(True)
        pts = torch.rand(n, wl.H, wl.L, wl.P, 2, device=dev, generator=g).to(dt).requires_grad_(True)
        att = torch.softmax(torch.randn(n, wl.H, wl.L * wl.P, device=dev, generator=g), -1)
        att = att.reshape(n, wl.H, wl.L, wl.P).to(dt).requires_grad_(True)
        shapes = torch.tensor(wl.levels, device=dev)

        def fwd():
            return row_sharded_multiscale_deformable_attention(value, shapes, pts, att, wl.padding_mode, wl.align_corners,
                                                               inputs_are_sharded=True, num_queries=wl.Q,
                                                               compute_only_as=(world, rank), overlap_chunks=chunks)

        def step():
            out = fwd()
            # (the gradient of this rank's rows only, as the backward slices it: rand_like of the whole result would
            #  charge the shard for N times its own rows)
            out.backward(out.detach())
            value.grad = pts.grad = att.grad = None

        def sync():
            if on_gpu:
                torch.cuda.synchronize()

        def timed(fn):
            for _ in range(warmup):
                fn()
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            sync()
            return (time.perf_counter() - t0) * 1e3 / steps

        with torch.no_grad():
            t_f = timed(fwd)
        t_fb = timed(step)
        plane = wl.I * wl.H * wl.D * (torch.finfo(dt).bits // 8)
        return {"rows": n, "fwd_ms": round(t_f, 4), "fwd_bwd_ms": round(t_fb, 4),
                "owners_sum_bytes": owners_sum_bytes(wl.B, wl.Q, world, rank, plane)}

    out = {"what": "rank 0 of an N-rank row-sharded job on ONE GPU, exchanges left out (compute_only_as): the speed-up "
                   "ceiling of the compute alone; speedup_ceiling = fwd_bwd t(1) / t_shard(N)"}
    strong = synth.WORKLOADS[strong_name]
    weak = synth.WORKLOADS[weak_name]
    for key, base, scale in (("strong_" + strong_name, strong, False), ("weak_" + weak_name, weak, True)):
        leg = {}
        one = time_shard(base, 1)
        leg["1"] = one
        for nn in ns:
            wl = dataclasses.replace(base, Q=base.Q * nn) if scale else base
            r = time_shard(wl, nn)  # in the pieces the operator itself picks (default_overlap_chunks: one, unless the
            #                         forward is long enough to hide a piece's exchange)
            r["speedup_ceiling"] = round(one["fwd_bwd_ms"] / r["fwd_bwd_ms"], 3)
            r["fwd_speedup_ceiling"] = round(one["fwd_ms"] / r["fwd_ms"], 3)
            r["ideal"] = 1 if scale else nn
            s = wl.elem_size
            r["pieces"] = default_overlap_chunks(wl.B * wl.Q, nn, 4 * wl.L * wl.P * wl.H * wl.D * s, wl.H * wl.D * s)
            r4 = time_shard(wl, nn, chunks=4)  # ... and what four overlapped pieces would cost in kernel time
            r["four_pieces"] = {"fwd_ms": r4["fwd_ms"], "fwd_bwd_ms": r4["fwd_bwd_ms"],
                                "speedup_ceiling": round(one["fwd_bwd_ms"] / r4["fwd_bwd_ms"], 3)}
            leg[str(nn)] = r
            if on_gpu:
                torch.cuda.empty_cache()
        out[key] = leg
    return out


_REAL_STDOUT = None  # file descriptor of the process's stdout once library output has been sent to stderr


def _library_output_to_stderr():
    """RCCL prints a version block ("RCCL version : ...", "Librccl path : ...") on STDOUT when its first communicator comes
    up.  Rank 0's stdout must carry ONE JSON line and nothing else, so in runs that initialise RCCL the process's fd 1 is
    pointed at stderr for everything — C libraries included — and the line is written to the saved descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    if _REAL_STDOUT is None:
        print(line, flush=True)
    else:
        sys.stdout.flush()
        os.write(_REAL_STDOUT, (line + "\n").encode())


class _StallGuard:
    """N>1 only.  A step that exchanges over RCCL is armed with a deadline; if it stalls, rank 0 prints the result
    measured so far (with config.stalled naming the step) and every rank leaves the process, so a wedged exchange
    costs the optional legs and not the bench line."""

    def __init__(self, rank):
        self.rank, self.result, self.timer, self.what = rank, None, None, ""
        self.have_result = False  # set on every rank once rank 0 holds a complete bench line

    def arm(self, seconds, what):
        import threading
        self.disarm()
        self.what = what
        self.timer = threading.Timer(seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def _fire(self):
        if self.rank == 0 and self.result is not None:
            self.result["config"]["stalled"] = self.what
            emit(json.dumps(self.result))
        sys.stderr.write(f"bench.py: rank {self.rank}: {self.what} stalled; leaving\n")
        sys.stderr.flush()
        os._exit(0 if self.have_result else 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c2_q10k")
    ap.add_argument("--spin-up-ms", type=float, default=200.0,
                    help="untimed steps for this long before the W warm-up steps (0: none); see spin_up()")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--xcd-map", type=int, default=None, help="override the blockIdx->(b,h) mapping (A/B runs)")
    ap.add_argument("--grad-value-sync", default="owners", choices=["owners", "all_reduce", "none"],
                    help="multi-GPU: how grad_value is combined (owners: among the ranks sharing a batch element)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and use the sharded code path even with one rank (self-test)")
    ap.add_argument("--no-strong-c5", action="store_true", help="skip the strong-scaling leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs (N=1 only)")
    ap.add_argument("--no-shard-compute", action="store_true",
                    help="skip the shard_compute_bound leg (rank 0's compute of an N = 2 / 4 / 8 job on one GPU)")
    ap.add_argument("--no-do-bench", action="store_true", help="skip the do_bench (cold / warm quantiles) leg (N=1 only)")
    ap.add_argument("--no-triton", action="store_true",
                    help="skip the Triton comparator leg (scripts/triton_comparator.py; N=1 only)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --device cpu: dry run of the N-rank control flow on host tensors")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT",
                    help="msda_set_option override for A/B runs, e.g. --opt value_path=2 --opt overlap=0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from msda_triton_amd import _lib, synth
    from msda_triton_amd.distributed import row_shard_bounds, row_sharded_multiscale_deformable_attention
    from msda_triton_amd.functional import KernelTimer, multiscale_deformable_attention

    SPIN_UP_MS = args.spin_up_ms
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    on_gpu = args.device == "cuda"
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback); --device cpu --backend gloo is "
                             "a dry run of the control flow only")
        if args.backend != "nccl":
            raise SystemExit("--device cuda goes with --backend nccl (RCCL)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        if args.backend != "gloo":
            raise SystemExit("--device cpu goes with --backend gloo")
        dev = torch.device("cpu")
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if on_gpu:
            _library_output_to_stderr()
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if on_gpu:
        if args.xcd_map is not None:
            _lib.set_option("xcd_map", args.xcd_map)
        for kv in args.opt:
            key, val = kv.split("=")
            _lib.set_option(key, int(val))

    wl = synth.WORKLOADS[args.workload]
    pm, ac = wl.padding_mode, wl.align_corners
    # Weak scaling: the global problem has B batch elements of world*Q queries; rank r owns B*Q contiguous rows of
    # the flattened (b, q) row space — whole batch elements while the ranks divide B.
    import dataclasses
    gwl = dataclasses.replace(wl, Q=wl.Q * world)  # (keeps the workload's sampling-point distribution, synth loc_mode)
    in_dt = None if on_gpu else torch.float32
    if use_dist:
        r0, r1 = row_shard_bounds(gwl.B * gwl.Q, world, rank)
        d = synth.make_inputs_torch(gwl, dev, seed=0, rows=(r0, r1), dtype=in_dt)
    else:
        d = synth.make_inputs_torch(gwl, dev, seed=0, dtype=in_dt)
    img, shapes = d["value"].requires_grad_(True), d["shapes"]
    pts, attn = d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)
    d.pop("grad_out", None)  # (the step draws its own, as the reference's benchmark does)

    exchange = {"chunks": None}  # None: the operator's default (one piece for these shapes); 1: one in-place all-gather; 4: pieces

    def op():
        if not use_dist:
            return multiscale_deformable_attention(img, shapes, pts, attn, pm, ac)
        return row_sharded_multiscale_deformable_attention(img, shapes, pts, attn, pm, ac, inputs_are_sharded=True,
                                                           num_queries=gwl.Q, grad_value_sync=args.grad_value_sync,
                                                           overlap_chunks=exchange["chunks"])

    def step():
        out = op()
        out.backward(torch.rand_like(out))
        img.grad = pts.grad = attn.grad = None

    def fwd_only():
        with torch.no_grad():
            op()

    def barrier():
        if on_gpu:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    def timed(fn, n, collect=True):
        import gc
        if collect:
            gc.collect()  # (tens of ms during which the GPU idles: the headline measurement collects BEFORE its warm-up)
        gc.disable()  # no collector pause inside the K steps (the loop allocates only tensors, freed by refcount)
        try:
            barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            barrier()
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        if use_dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    guard = _StallGuard(rank)
    failed_legs = []

    def spin_up(fn):
        """Untimed steps for SPIN_UP_MS before the W warm-up steps: the reference's benchmark warms up for >= 100 ms
        before it times anything (scripts/benchmark.py:52-54, triton.testing.do_bench), and this GPU needs that long:
        after an idle spell (process start-up, a collector pause, a barrier) the same step takes 0.35 ms for the first
        ~20 ms of work and settles at 0.315 ms after ~50 (tools/step_ramp.py) — a W = 5 / K = 20 window started cold reads
        6-10 % above the steady state, which is what made the driver's end-of-round figure differ from this file's default
        run in rounds 1-4.  Every rank runs the same number of steps (they carry collectives)."""
        if not on_gpu or SPIN_UP_MS <= 0:
            return 0
        for _ in range(3):  # (the very first calls load code objects and grow the allocator: not a step time)
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        barrier()
        per = (time.perf_counter() - t0) / 5
        if use_dist:
            t = torch.tensor([per], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            per = float(t.item())
        n = int(min(2000, max(0.0, SPIN_UP_MS / 1e3 / max(per, 1e-6))))
        for _ in range(n):
            fn()
        return n + 8

    spin = {"steps": 0}

    def measure():
        import gc
        gc.collect()
        spin["steps"] = spin_up(step)
        for _ in range(args.warmup):
            step()
        # ---- the timed region: exactly K steps of the un-instrumented public API, straight behind the warm-up ----
        ms = timed(step, args.steps, collect=False) * 1e3 / args.steps
        for _ in range(max(3, args.warmup // 2)):
            fwd_only()
        return ms, timed(fwd_only, args.steps, collect=False) * 1e3 / args.steps

    def cold_window():
        """The protocol of rounds 1-4 (and of any caller that times a short burst on an idle GPU): the device idles,
        W warm-up steps, a collector pause + barrier, then the K timed steps — no spin-up.  Reported NEXT to the
        steady-state `ms_per_step` so that records from before and after round 5's spin-up change compare."""
        import gc
        barrier()
        time.sleep(0.3)
        for _ in range(args.warmup):
            step()
        gc.collect()
        return timed(step, args.steps, collect=True) * 1e3 / args.steps

    exchange_ms = None
    if use_dist and world > 1:
        # Two ways to exchange the output rows: one in-place all-gather after the kernels, or grouped point-to-point
        # pieces overlapped with the compute.  The second has only ever run over gloo in the build container (no
        # multi-GPU box there), so the all-gather is measured first and the piece-wise exchange runs under a stall
        # guard: if it raises or stalls, the all-gather figure stands.  Both times are reported; `value` is the
        # faster one (config.exchange says which).
        exchange["chunks"] = 1
        guard.arm(600, "weak-scaling leg (all-gather exchange)")
        ms_step, ms_fwd = measure()
        guard.disarm()
        exchange_ms = {"all_gather": {"fwd_bwd_ms": ms_step, "fwd_ms": ms_fwd}}
    else:
        ms_step, ms_fwd = measure()
    # ---- the same K steps again with per-launch HIP events (KernelTimer splits the backward into one C-ABI call
    #      per kernel group, so its two halves run back to back here instead of concurrently) ----
    kern = {}
    peak_mem = peak_ref = launch_info = None
    if on_gpu:
        for _ in range(args.warmup):
            step()
        with KernelTimer() as kt:
            timed(step, args.steps, collect=False)  # (no collector pause in front: the GPU stays in its steady state)
        kern = kt.summary()
        torch.cuda.synchronize()
        # ... and once more with the library's own event pair around every single kernel (msda_profile_read)
        single = {}
        try:
            _lib.set_option("profile", 1)
            for _ in range(args.warmup):  # (the first profiled launches create their event pairs: not measured)
                step()
            torch.cuda.synchronize()
            _lib.profile_read()
            timed(step, args.steps, collect=False)
            single = _lib.profile_read()
        finally:
            _lib.set_option("profile", 0)
        torch.cuda.reset_peak_memory_stats(dev)
        step()
        torch.cuda.synchronize()
        peak_mem = torch.cuda.max_memory_allocated(dev)
        # ... and by the reference's recipe (scripts/benchmark.py:158-172; README.md:20 quotes 166.14 MB): the inputs stay
        # resident, start = memory_allocated(), one fwd+bwd, max_memory_allocated() - start — what a step allocates on top
        peak_ref, reps = 0.0, 10
        for _ in range(reps):
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats(dev)
            start_mem = torch.cuda.memory_allocated(dev)
            step()
            torch.cuda.synchronize()
            peak_ref += (torch.cuda.max_memory_allocated(dev) - start_mem) / 1e6 / reps
        fwd_only()
        launch_info = _lib.last_launch_info()
    ms_cold = cold_window()

    if rank == 0:
        kernels = kernel_table(wl, kern, load_traffic(args.workload))
        sum_kernel_us = sum(k["avg_us"] for k in kernels.values())
        weak_note = ("" if world == 1 else
                     f" — WEAK scaling: every rank keeps {wl.Q} queries per batch element, the global problem has "
                     f"Q = {wl.Q} x {world} = {gwl.Q}")
        result = {
            "metric": "MSDA fwd+bwd @10k queries per GPU (fwd_ms / fwd_bwd_ms alongside), 1 MI355X per rank" + weak_note,
            "value": wl.B * wl.Q * world / (ms_step * 1e-3),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "ms_per_step_cold_window": ms_cold,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"float32": "f32", "float16": "f16", "bfloat16": "bf16", "float64": "f64"}[wl.dtype],
            "data": "synthetic",
            "config": {"workload": f"{wl.name}: B={wl.B} Q={wl.Q}/rank H={wl.H} D={wl.D} L={wl.L} "
                                   f"levels={list(wl.levels)} P={wl.P} {wl.dtype} {pm} align_corners={ac}",
                       "global_queries": gwl.Q,
                       "scaling_note": "weak: global Q = per-rank Q x N (the metric's 10k queries are per GPU); the "
                                       "strong_scaling leg keeps the global problem fixed",
                       "parallelism": f"row-shard x{world} (B*Q rows per rank, kernels write in place, in-place exchange "
                                      f"overlapped with compute{', grad_value ' + args.grad_value_sync if use_dist else ''})",
                       "step": "public autograd API: fwd + backward(rand_like(out)) + grad reset",
                       "spin_up": {"ms": SPIN_UP_MS, "steps": spin["steps"],
                                   "what": "untimed steps in front of the W warm-up steps (the reference benchmark's own >= 100 ms "
                                           "do_bench warm-up, scripts/benchmark.py:52-54): a GPU that has just idled runs the same "
                                           "step 6-10 % slower for its first ~50 ms (tools/step_ramp.py); --spin-up-ms 0 turns it off"},
                       "exchange": ("none (one rank)" if world == 1 else
                                    "one in-place all-gather" if exchange["chunks"] == 1 else
                                    "grouped point-to-point pieces overlapped with compute"),
                       "backend": args.backend, "device": args.device},
            "fwd_ms": ms_fwd,
            "fwd_bwd_ms": ms_step,
            "sum_kernel_us_fwd_bwd": round(sum_kernel_us, 2),
            "alg_fwd_bwd_GBs": round((wl.alg_fwd_bytes + wl.alg_bwd_bytes) / (ms_step * 1e-3) / 1e9, 1),
            "effective_gather_GBs_fwd": round(wl.gather_fwd_bytes / (kernels["msda_fwd"]["avg_us"] * 1e-6) / 1e9, 1)
            if "msda_fwd" in kernels else None,
            "reference_readme_rtx2060_ms": README_RTX2060_MS,
            "speedup_vs_reference_readme": {"fwd": README_RTX2060_MS["fwd"] / ms_fwd,
                                            "fwd_bwd": README_RTX2060_MS["fwd_bwd"] / ms_step},
            "kernels": kernels,
            "peak_mem_MB": round(peak_mem / 1e6, 1) if peak_mem is not None else None,
            "peak_mem_reference_recipe_MB": round(peak_ref, 1) if peak_ref is not None else None,
            "peak_mem_note": "peak_mem_MB: max_memory_allocated over a step, inputs included; peak_mem_reference_recipe_MB: "
                             "the reference's recipe (scripts/benchmark.py:158-172) — what one fwd+bwd allocates ON TOP of "
                             "its resident inputs (result, rand_like gradient, three gradients, workspace); the reference's "
                             "README quotes 166.14 MB for it",
        }
        if on_gpu and single:
            salg = single_kernel_alg_bytes(wl)
            result["single_kernels"] = {
                name: {"launches": n, "avg_us": round(us, 2),
                       **({"alg_bytes": salg[name], "achieved_GBs": round(salg[name] / (us * 1e-6) / 1e9, 1),
                           "frac_of_hbm_peak": round(salg[name] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)} if name in salg else {})}
                for name, (n, us) in sorted(single.items(), key=lambda kv: -kv[1][1])}
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]["avg_us"])
            sk = result.get("single_kernels", {})
            dom_single = max(sk, key=lambda k: sk[k]["avg_us"]) if sk else None
            fwd_us = sk.get("msda_fwd_kernel", {}).get("avg_us") or kernels.get("msda_fwd", {}).get("avg_us")
            result["roofline"] = {
                "kernel": dom + (" (grad_value: count, scan, place, value_gather, value_finish — or the "
                                 "single-launch kernel on small problems; timed as one C-ABI call)"
                                 if dom == "msda_bwd_value" else ""),
                "bound": "hbm", "achieved": kernels[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kernels[dom]["frac_of_hbm_peak"], "traffic": kernels[dom]["traffic"],
                "traffic_source": "profiles/hbm_traffic.json: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of this "
                                  "bench, gfx950 corrections applied; not re-measured in this run",
                "timing": "HIP events around every launch, same K steps repeated right after the timed region",
                # the line's `kernel` is a launch GROUP (one C-ABI call); the slowest SINGLE kernel, with its own figure:
                "dominant_single_kernel": ({"kernel": dom_single, **sk[dom_single]} if dom_single else None),
                # the forward's gather as the hardware sees it: logical row bytes per second, and as a fraction of
                # the vector L1 rate (64 B/clk/CU x 256 CUs x 2.4 GHz) — "bound by requests, not HBM bytes"
                "effective_gather_TBs": round(wl.gather_fwd_bytes / (fwd_us * 1e-6) / 1e12, 2) if fwd_us else None,
                "traffic_caveat": "the 2x FETCH_SIZE correction is calibrated for 16-byte-per-lane streaming reads only: "
                                  "for the scatter / gather kernels `traffic` is an upper-ish bound (profiles/hbm_traffic.json)"}
            if fwd_us and launch_info:
                # the forward's gather split by where its rows come from: through the vector L1 (64 B/clk/CU) or out of the
                # workgroup's LDS copy of the coarse levels (ds_read_b128, 256 B/clk/CU) — two fractions of two different
                # peaks, each below 1 (rounds 4-5 divided ALL rows by the L1 rate and read 1.015 once half of them had moved)
                f_lds = lds_row_fraction(wl, launch_info)
                rows_bytes = wl.gather_fwd_bytes / (fwd_us * 1e-6) / 1e9
                result["roofline"].update({
                    "fwd_rows_from_lds_frac": round(f_lds, 3),
                    "gather_memory_side_frac_of_l1_rate": round(rows_bytes * (1 - f_lds) / L1_PEAK_GBS, 3),
                    "gather_lds_side_frac_of_lds_rate": round(rows_bytes * f_lds / LDS_PEAK_GBS, 3),
                    "fwd_launch": launch_info})
        if on_gpu:
            result["options"] = {k: _lib.get_option(k) for k in ("xcd_map", "value_path", "overlap", "ws_passes")}
    else:
        result = None
    guard.result, guard.have_result = result, True

    # ---- optional legs: none of them may cost rank 0 its JSON line ----
    def optional(key, fn, collective):
        """Run one optional leg; an exception is recorded under `key` (collective legs are attempted on every rank: an
        error in a collective is raised on all of them, a stall trips the guard)."""
        try:
            if os.environ.get("MSDA_BENCH_INJECT_FAIL") == key:  # test hook (tests/test_distributed_cpu.py)
                raise RuntimeError(f"injected failure in optional leg {key}")
            out = fn()
        except Exception as e:  # noqa: BLE001
            out = {"error": repr(e)[:300]}
            failed_legs.append(key)
        if rank == 0 and out is not None:
            result[key] = out
        return out

    pieces_failed = False
    if exchange_ms is not None:
        guard.arm(240, "piece-wise exchange (weak-scaling leg)")
        # (four pieces, forced: the operator's own default is ONE piece for these shapes — default_overlap_chunks: the
        #  forward cannot hide a piece's exchange — so this leg is what says whether that rule holds on real links)
        exchange["chunks"] = 4
        try:
            ms2, fwd2 = measure()
            exchange_ms["pieces"] = {"fwd_bwd_ms": ms2, "fwd_ms": fwd2}
        except Exception as e:  # every rank sees the same RCCL error; a rank left waiting trips the guard
            ms2 = None
            pieces_failed = True
            failed_legs.append("exchange_ms.pieces")
            exchange_ms["pieces"] = {"error": repr(e)[:300]}
        guard.disarm()
        if ms2 is not None and ms2 < ms_step:  # the all-reduced max over ranks: the same decision everywhere
            ms_step, ms_fwd = ms2, fwd2
        else:
            exchange["chunks"] = 1
        if rank == 0:
            result.update(value=wl.B * wl.Q * world / (ms_step * 1e-3), ms_per_step=ms_step, fwd_bwd_ms=ms_step,
                          fwd_ms=ms_fwd, exchange_ms=exchange_ms,
                          alg_fwd_bwd_GBs=round((wl.alg_fwd_bytes + wl.alg_bwd_bytes) / (ms_step * 1e-3) / 1e9, 1),
                          speedup_vs_reference_readme={"fwd": README_RTX2060_MS["fwd"] / ms_fwd,
                                                       "fwd_bwd": README_RTX2060_MS["fwd_bwd"] / ms_step})
            result["config"]["exchange"] = ("one in-place all-gather" if exchange["chunks"] == 1 else
                                            "grouped point-to-point pieces overlapped with compute")
    if world == 1 and on_gpu and rank == 0:
        if not args.no_do_bench:
            def leg_do_bench():
                return {"recipe": "triton.testing.do_bench semantics (reference scripts/benchmark.py:38,52-54): >=100 ms "
                                  "warm-up, >=1 s of repetitions, one HIP-event pair per repetition, median + p20/p80; "
                                  "cold = a 512 MiB buffer rewritten before every repetition (L2 + Infinity Cache flushed), "
                                  "warm = back to back",
                        "fwd": {"cold": do_bench(fwd_only, dev, True), "warm": do_bench(fwd_only, dev, False)},
                        "fwd_bwd": {"cold": do_bench(step, dev, True), "warm": do_bench(step, dev, False)}}
            optional("do_bench", leg_do_bench, False)
        if not args.no_cpu_baseline:
            optional("cpu_baseline", lambda: cpu_baseline(wl), False)
    strong_wl = "c5_stress" if on_gpu else "dryrun_strong"
    run_strong = not args.no_strong_c5 and args.workload in ("c2_q10k", "dryrun")
    if run_strong and pieces_failed:
        # the communicator that just raised is not reused (its error state would only raise again)
        if rank == 0:
            result["strong_scaling_c5"] = {"skipped": "the piece-wise exchange failed on this communicator"}
        run_strong = False
    if run_strong:
        del img, pts, attn, d
        if on_gpu:
            torch.cuda.empty_cache()
        if world > 1:
            guard.arm(300, "strong-scaling leg")
        optional("strong_scaling_c5",
                 lambda: strong_scaling_leg(strong_wl, dev, world, rank, use_dist, exchange["chunks"]), True)
        guard.disarm()
        if on_gpu:
            torch.cuda.empty_cache()
    if world == 1 and rank == 0 and not args.no_shard_compute and args.workload in ("c2_q10k", "dryrun"):
        optional("shard_compute_bound",
                 lambda: shard_compute_leg(dev, "c5_stress" if on_gpu else "dryrun_strong", args.workload), False)
        if on_gpu:
            torch.cuda.empty_cache()
    if world == 1 and on_gpu and rank == 0 and not args.no_configs and args.workload == "c2_q10k":
        def leg_configs():
            out = {}
            for name in ("c1_readme", "c2_q1k", "c2_q5k", "c2_q10k_zeros", "c3_ddetr_enc", "c3_ddetr_enc_local",
                         "c4_gdino_dec"):
                try:
                    out[name] = bench_config(name, dev)
                except Exception as e:  # noqa: BLE001
                    out[name] = {"error": repr(e)[:300]}
                    failed_legs.append(f"configs.{name}")
                torch.cuda.empty_cache()
            return out
        optional("configs", leg_configs, False)
        optional("decoder_real_pyramid", lambda: real_pyramid_leg(dev), False)
        optional("padded_value_rows", lambda: padded_rows_leg(args.workload, dev), False)
    if world == 1 and on_gpu and rank == 0 and not args.no_triton and args.workload == "c2_q10k":
        result["triton_comparator"] = triton_comparator_leg(args.workload, dev)
    if rank == 0:
        if failed_legs:
            result["failed_legs"] = failed_legs
        emit(json.dumps(result))
    if use_dist:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 (a communicator in error state: the line is out, just leave)
            pass
    if failed_legs:
        sys.exit(1)


if __name__ == "__main__":
    main()
