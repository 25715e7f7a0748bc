"""dev tool: the HIP operator next to the Triton comparator (scripts/triton_comparator.py) over a shape matrix — looks
for shapes where the hand-written kernels lose.  fp32; prints fwd / fwd+bwd ms of both and the ratios."""
import importlib.util, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from msda_triton_amd import synth
spec = importlib.util.spec_from_file_location("tc", os.path.join(ROOT, "scripts", "triton_comparator.py"))
tc = importlib.util.module_from_spec(spec); spec.loader.exec_module(tc)
PYR = ((64, 64), (32, 32), (16, 16), (8, 8))
cases = [
    ("D16", dict(B=4, Q=5000, H=8, D=16, levels=PYR, P=4)),
    ("D64", dict(B=4, Q=5000, H=8, D=64, levels=PYR, P=4)),
    ("D128", dict(B=2, Q=5000, H=8, D=128, levels=PYR, P=4)),
    ("D256", dict(B=2, Q=2000, H=4, D=256, levels=PYR, P=4)),
    ("D24", dict(B=4, Q=5000, H=8, D=24, levels=PYR, P=4)),
    ("P8", dict(B=4, Q=5000, H=8, D=32, levels=PYR, P=8)),
    ("P1", dict(B=4, Q=5000, H=8, D=32, levels=PYR, P=1)),
    ("L1", dict(B=4, Q=5000, H=8, D=32, levels=((64, 64),), P=4)),
    ("L1big", dict(B=4, Q=5000, H=8, D=32, levels=((200, 200),), P=4)),
    ("H1", dict(B=4, Q=5000, H=1, D=32, levels=PYR, P=4)),
    ("H16", dict(B=2, Q=5000, H=16, D=32, levels=PYR, P=4)),
    ("B1", dict(B=1, Q=10000, H=8, D=32, levels=PYR, P=4)),
    ("B32q300", dict(B=32, Q=300, H=8, D=32, levels=PYR, P=4)),
    ("realdec", dict(B=8, Q=900, H=8, D=32, levels=((100, 134), (50, 67), (25, 34), (13, 17)), P=4)),
    ("realenc", dict(B=2, Q=17821, H=8, D=32, levels=((100, 134), (50, 67), (25, 34), (13, 17)), P=4)),
    ("Q100", dict(B=2, Q=100, H=8, D=32, levels=PYR, P=4)),
    ("Q10", dict(B=2, Q=10, H=8, D=32, levels=PYR, P=4)),
]
dev = torch.device("cuda")
rows = {}
for name, kw in cases:
    for pm, ac in (("border", True), ("zeros", False)):
        wl = synth.Workload(name, dtype="float32", padding_mode=pm, align_corners=ac, **kw)
        try:
            r = tc.compare(wl, dev, steps=40, warmup=8)
            key = f"{name}/{pm}"
            rows[key] = r
            print(f"{key:18s} fwd hip {r['hip']['fwd_ms']:.4f} triton {r['triton']['fwd_ms']:.4f} ({r['hip_speedup']['fwd_ms']:.2f}x) | "
                  f"fwd+bwd hip {r['hip']['fwd_bwd_ms']:.4f} triton(relaxed) {r['triton_relaxed_atomics']['fwd_bwd_ms']:.4f} "
                  f"({r['hip_speedup']['fwd_bwd_ms_vs_relaxed_atomics']:.2f}x) default-atomics {r['triton']['fwd_bwd_ms']:.3f} | "
                  f"max diff out {r['max_abs_diff']['out']:.1e} gv {r['max_abs_diff']['grad_value']:.1e}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"{name}/{pm}: ERROR {e!r}"[:300], flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
