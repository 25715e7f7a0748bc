"""dev tool: cold-cache forward latency at small Q (the reference benchmark's recipe, scripts/benchmark_sweep.py do_bench)
for option strings alternated INSIDE one process (boxes and even runs differ by more than the effects looked for):
    python tools/small_q_cold.py [--triton] [--floor] [--warm] [--reps N] [--flush MiB] [--spin ms] Q [Q ...] -- opt=val[,opt=val] [opt=val ...]
("-" = defaults).  --triton: the comparator takes its turn in every round; --flush 1024: the Infinity Cache is flushed too
(256, the recipe's figure, leaves an undefined share of the pyramid there: HISTORY.md 9); --reps: rounds of the option list."""
import importlib.util
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msda_triton_amd import _lib, multiscale_deformable_attention  # noqa: E402

spec = importlib.util.spec_from_file_location("sweep", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts", "benchmark_sweep.py"))
sweep = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sweep)

args = sys.argv[1:]
want_triton = "--triton" in args
if "--warm" in args:  # back-to-back instead of the flushed cache
    torch.Tensor.zero_ = lambda self: self
reps = 2
spin = 0.0
if "--spin" in args:  # ms of the sweep's spin-up in front of every measurement (default here: none)
    i = args.index("--spin")
    spin = float(args[i + 1])
    del args[i:i + 2]
if "--flush" in args:  # MiB zeroed before every repetition (256: the recipe; 1024: also the Infinity Cache)
    i = args.index("--flush")
    sweep.FLUSH_MIB = int(args[i + 1])
    del args[i:i + 2]
if "--reps" in args:  # repetitions of the whole option list (the first measurement after make_inputs tends to read high)
    i = args.index("--reps")
    reps = int(args[i + 1])
    del args[i:i + 2]
args = [a for a in args if a not in ("--triton", "--warm", "--floor")]
split = args.index("--") if "--" in args else len(args)
qs = [int(a) for a in args[:split]] or [10, 100, 300, 900, 1000]
opts = args[split + 1:] or ["-"]
tc = None
if want_triton:
    s2 = importlib.util.spec_from_file_location("tc", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts", "triton_comparator.py"))
    tc = importlib.util.module_from_spec(s2)
    s2.loader.exec_module(tc)


def apply(o, defaults):
    for k, v in defaults.items():
        _lib.set_option(k, v)
    if o != "-":
        for kv in o.split(","):
            k, v = kv.split("=")
            _lib.set_option(k, int(v))


keys = sorted({kv.split("=")[0] for o in opts if o != "-" for kv in o.split(",")})
defaults = {k: _lib.get_option(k) for k in keys}
for N in qs:
    img, shapes, pts, att = sweep.make_inputs(N, False)

    def fwd():
        with torch.no_grad():
            multiscale_deformable_attention(img, shapes, pts, att, "border", True)

    def tfwd():
        with torch.no_grad():
            tc.triton_comparator_msda(img, shapes, pts, att, "border", True)

    have_triton = tc is not None and tc.HAVE_TRITON
    if have_triton:
        tfwd()  # (compile + autotune before anything is measured)
    line = {}
    for rep in range(reps):  # the comparator takes its turn in every round, like an option
        for o in opts:
            apply(o, defaults)
            line.setdefault(o, []).append(sweep.do_bench(fwd, warmup_ms=30.0, rep_ms=300.0, spin_ms=spin)[0] * 1e3)
        apply("-", defaults)
        if have_triton:
            line.setdefault("triton", []).append(sweep.do_bench(tfwd, warmup_ms=30.0, rep_ms=300.0, spin_ms=spin)[0] * 1e3)
    if "--floor" in sys.argv:  # a kernel that does nothing, by the same recipe: what of the figures is the launch itself
        tiny = torch.zeros(64, device=img.device)
        line["floor(fill 64 floats)"] = [sweep.do_bench(lambda: tiny.fill_(1.0), warmup_ms=30.0, rep_ms=300.0, spin_ms=spin)[0] * 1e3 for _ in range(2)]
    print("Q=%5d " % N + "  ".join("%s: %s us" % (k, "/".join("%.2f" % x for x in v)) for k, v in line.items()), flush=True)
