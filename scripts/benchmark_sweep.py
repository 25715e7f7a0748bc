#!/usr/bin/env python3
"""Runtime / memory sweep over the number of queries — the measurement the reference publishes
(/root/reference/scripts/benchmark.py:10-180, README.md:6-22): forward ms, forward+backward ms and peak
memory for Q in {10, 100, 300, 900, 1000, 10000}; B=4, H=8, C=32, L=4 levels 64..8, P=4, fp32,
border / align_corners=True; providers: the HIP kernels, the plain-PyTorch formulation on the GPU and the Triton
comparator written for this repo (scripts/triton_comparator.py — not the reference's kernel).

Timing follows triton.testing.do_bench's recipe (which the reference uses): ~100 ms warm-up, ~1 s of
repetitions, one HIP-event pair per repetition, an L2-sized buffer zeroed before each, median and the
20 / 80 % quantiles — behind 0.7 s of the same loop unmeasured (--spin-up-ms; see do_bench).  Results go to outputs/benchmark_results/*.csv.

The three plots the reference ships (assets/images/msda *.png; benchmark.py:177-180 saves them next to the CSV) are
written as PNGs beside the CSV: forward ms, forward+backward ms and peak memory over the number of queries.

    python scripts/benchmark_sweep.py [--queries 10 100 ...] [--no-native] [--no-plots]
"""
import argparse
import csv
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msda_triton_amd import multiscale_deformable_attention, native_multiscale_deformable_attention  # noqa: E402

SHAPES = [(64, 64), (32, 32), (16, 16), (8, 8)]
B, H, C, P = 4, 8, 32, 4
SPIN_UP_MS = 700.0  # (--spin-up-ms)
# Bytes zeroed before every repetition (--flush-mib).  256 MiB is triton.testing.do_bench's figure — enough for the L2s,
# but exactly the size of the 256 MB Infinity Cache behind them: how much of the inputs survives there depends on what
# ran before (tools/cold_state_probe.py: the same call reads 25.3 us on fresh inputs, 21.6 after a plain img.sum(), 20.4
# after one call in another block order, 25.7 again after 1 GiB of zeros).  1024 makes "cold" mean HBM for everybody.
FLUSH_MIB = 256


def do_bench(fn, warmup_ms=100.0, rep_ms=1000.0, spin_ms=None):
    import time
    fn()
    torch.cuda.synchronize()
    flush = torch.empty(FLUSH_MIB << 20, dtype=torch.int8, device="cuda")
    # Spin-up (every provider alike): the measured loop itself — flush, call — for SPIN_UP_MS of wall time first.  A GPU
    # that has just idled (the inputs were being made, the previous result reduced on the host) serves the first ~0.5 s
    # of this loop 20-25 % slower than the rest (tools/small_q_cold.py --reps 3: the first option after make_inputs reads
    # 22.6 us at Q = 900, every later one 17.3), so whoever was measured first looked slow.  0 turns it off.
    t_end = time.perf_counter() + (SPIN_UP_MS if spin_ms is None else spin_ms) * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(20):
            flush.zero_()
            fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        flush.zero_()
        fn()
    e.record()
    torch.cuda.synchronize()
    est = max(s.elapsed_time(e) / 5, 1e-3)
    n_warm, n_rep = max(1, int(warmup_ms / est)), max(5, min(2000, int(rep_ms / est)))
    for _ in range(n_warm):
        fn()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(n_rep)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(n_rep)]
    for i in range(n_rep):
        flush.zero_()
        starts[i].record()
        fn()
        ends[i].record()
    torch.cuda.synchronize()
    t = torch.tensor([a.elapsed_time(b) for a, b in zip(starts, ends)])
    q = torch.quantile(t, torch.tensor([0.5, 0.2, 0.8]))
    return [float(x) for x in q]


def make_inputs(N, requires_grad):
    L = len(SHAPES)
    I = sum(h * w for h, w in SHAPES)  # noqa: E741
    img = torch.randn(B, I, H, C, device="cuda", requires_grad=requires_grad)
    shapes = torch.tensor(SHAPES, device="cuda")
    pts = torch.rand(B, N, H, L, P, 2, device="cuda", requires_grad=requires_grad)
    att = torch.softmax(torch.randn(B, N, H, L, P, device="cuda"), dim=-1).requires_grad_(requires_grad)
    return img, shapes, pts, att


def main():
    global SPIN_UP_MS, FLUSH_MIB
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, nargs="+", default=[10, 100, 300, 900, 1000, 10000])
    ap.add_argument("--no-native", action="store_true")
    ap.add_argument("--no-plots", action="store_true")
    ap.add_argument("--no-triton", action="store_true", help="leave out the Triton comparator (scripts/triton_comparator.py)")
    ap.add_argument("--out", default="outputs/benchmark_results")
    ap.add_argument("--fwd-only", action="store_true", help="forward timings only (development: the small-Q latency work)")
    ap.add_argument("--flush-mib", type=int, default=FLUSH_MIB,
                    help="MiB zeroed before every repetition: 256 = the reference recipe (L2-cold, Infinity Cache undefined), 1024 = HBM-cold")
    ap.add_argument("--spin-up-ms", type=float, default=SPIN_UP_MS,
                    help="wall time of the flush + call loop run before every measurement (do_bench); 0: none")
    args = ap.parse_args()
    SPIN_UP_MS = args.spin_up_ms
    FLUSH_MIB = args.flush_mib
    providers = {"hip": multiscale_deformable_attention}
    if not args.no_native:
        providers["torch"] = native_multiscale_deformable_attention
    if not args.no_triton:  # the builder-authored Triton comparator (NOT the reference's kernel), scripts/triton_comparator.py
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location(
                "msda_triton_comparator", os.path.join(os.path.dirname(os.path.abspath(__file__)), "triton_comparator.py"))
            tc = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(tc)
            if tc.HAVE_TRITON:
                providers["triton"] = tc.triton_comparator_msda
        except Exception as e:  # noqa: BLE001
            print("no Triton comparator:", repr(e)[:200])
    os.makedirs(args.out, exist_ok=True)
    rows = []
    for N in args.queries:
        for name, op in providers.items():
            img, shapes, pts, att = make_inputs(N, False)

            def fwd():
                with torch.no_grad():
                    op(img, shapes, pts, att, "border", True)

            f = do_bench(fwd)
            if args.fwd_only:
                print(dict(num_queries=N, provider=name, fwd_ms=round(f[0], 5), p20=round(f[1], 5), p80=round(f[2], 5)), flush=True)
                continue
            img, shapes, pts, att = make_inputs(N, True)

            def fwdbwd():
                out = op(img, shapes, pts, att, "border", True)
                out.backward(torch.rand_like(out))
                img.grad = pts.grad = att.grad = None

            fb = do_bench(fwdbwd)
            # Peak memory by the reference's recipe (scripts/benchmark.py:158-172): the inputs stay alive, 10 warm-up
            # steps, then per repetition reset the peak, start = memory_allocated(), one fwd+bwd, max_memory_allocated()
            # - start, in MB of 1e6 bytes, averaged.  It counts what a step allocates ON TOP of its resident inputs —
            # the result, rand_like's gradient, the three gradients, and any workspace (README.md:20: 166.14 MB for
            # the reference = exactly those five tensors at Q = 10 000).  Rounds 1-5 of this script reported another
            # figure under the same name — the peak over a freshly made set of inputs INCLUDING them, in MiB; it is
            # kept as peak_mem_incl_inputs_MiB.
            del img, shapes, pts, att
            import gc
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            base = torch.cuda.memory_allocated()
            img, shapes, pts, att = make_inputs(N, True)
            torch.cuda.reset_peak_memory_stats()
            fwdbwd()
            torch.cuda.synchronize()
            mem_incl = (torch.cuda.max_memory_allocated() - base) / 2**20
            for _ in range(10):
                fwdbwd()
            mem, reps = 0.0, 20
            for _ in range(reps):
                torch.cuda.synchronize()
                torch.cuda.reset_peak_memory_stats()
                start = torch.cuda.memory_allocated()
                fwdbwd()
                torch.cuda.synchronize()
                mem += (torch.cuda.max_memory_allocated() - start) / 1e6
            mem /= reps
            rows.append(dict(num_queries=N, provider=name, fwd_ms=f[0], fwd_ms_p20=f[1], fwd_ms_p80=f[2],
                             fwdbwd_ms=fb[0], fwdbwd_ms_p20=fb[1], fwdbwd_ms_p80=fb[2], peak_mem_MB=mem,
                             peak_mem_incl_inputs_MiB=mem_incl))
            print(rows[-1], flush=True)
    if args.fwd_only:
        return
    path = os.path.join(args.out, "msda_sweep.csv")
    with open(path, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    print("wrote", path)
    if not args.no_plots:
        plot(rows, args.out)


def plot(rows, out_dir):
    """One PNG per measured quantity, a line per provider, log-log axes (reference: triton.testing.perf_report plots)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    labels = {"hip": "HIP kernels (this package, MI355X)", "torch": "plain PyTorch on the same GPU",
              "triton": "Triton comparator written here (atomics as the reference calls them)"}
    for key, title, fname in (("fwd_ms", "msda fwd runtime (ms)", "msda_fwd_runtime_ms.png"),
                              ("fwdbwd_ms", "msda fwd+bwd runtime (ms)", "msda_fwd_bwd_runtime_ms.png"),
                              ("peak_mem_MB", "msda memory consumption (MB)", "msda_memory_consumption_MB.png")):
        fig, ax = plt.subplots(figsize=(6, 4))
        for prov in sorted({r["provider"] for r in rows}):
            pts = sorted((r["num_queries"], r[key]) for r in rows if r["provider"] == prov)
            ax.plot([p[0] for p in pts], [p[1] for p in pts], marker="o", label=labels.get(prov, prov))
            if key != "peak_mem_MB":
                lo = [r[key + "_p20"] for r in sorted(rows, key=lambda r: r["num_queries"]) if r["provider"] == prov]
                hi = [r[key + "_p80"] for r in sorted(rows, key=lambda r: r["num_queries"]) if r["provider"] == prov]
                ax.fill_between([p[0] for p in pts], lo, hi, alpha=0.15)
        ax.set_xscale("log")
        ax.set_yscale("log")
        ax.set_xlabel("num_queries")
        ax.set_ylabel(title)
        ax.set_title(title)
        ax.grid(True, which="both", alpha=0.3)
        ax.legend()
        fig.tight_layout()
        fig.savefig(os.path.join(out_dir, fname), dpi=110)
        plt.close(fig)
        print("wrote", os.path.join(out_dir, fname))


if __name__ == "__main__":
    main()
