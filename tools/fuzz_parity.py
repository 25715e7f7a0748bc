#!/usr/bin/env python3
"""Long differential fuzz of the HIP operator against the C oracle (fp64 cases at 1e-8, fp32 at the test suite's
tolerances): wider than tests/test_gpu_parity.py::test_random_shapes_against_oracle — more queries (cells that span
many gather windows and workgroups), 1 x N levels, hot spots, far out-of-range coordinates, both grad_value paths,
both place passes, query rounds.  Usage: fuzz_parity.py [seconds] [first_seed]; prints one line per failure and a
summary; exit status 1 if anything failed.  fuzz_parity.py --repro SEED [TIMES] [option=value ...] repeats one case."""
import json
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import test_gpu_parity as tp
from msda_triton_amd import _lib
from oracle import msda_oracle


def make_case(seed):
    """-> (inputs, padding mode, align_corners, torch dtype, library options, description) of fuzz case `seed`"""
    rng = np.random.default_rng(77000 + seed)
    kind = seed % 6
    B, H = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    D = int(rng.choice([1, 2, 3, 7, 8, 16, 24, 32, 40, 64, 80]))
    L, P = int(rng.integers(1, 7)), int(rng.integers(1, 9))
    Q = int(rng.choice([1, 2, 17, 64, 200, 700, 1500, 3000])) if kind != 5 else int(rng.integers(1, 60))
    big = int(rng.choice([6, 12, 25, 40]))
    levels = [(int(rng.integers(1, big + 1)), int(rng.integers(1, big + 1))) for _ in range(L)]
    if kind == 1:
        levels[int(rng.integers(0, L))] = (1, int(rng.integers(1, 50)))  # a one-row level
    if kind == 2:
        levels[int(rng.integers(0, L))] = (int(rng.integers(1, 50)), 1)  # a one-column level
    lo, hi = [(-0.4, 1.4), (0.0, 1.0), (0.45, 0.55), (-30.0, 31.0), (0.0, 1.0), (-0.1, 1.1)][kind]
    pm, ac = tp.MODES[int(rng.integers(0, len(tp.MODES)))]
    f64 = bool(rng.integers(0, 2))
    # keep the oracle's work bounded: samples * D
    while B * Q * H * L * P * D > 6e7 and Q > 1:
        Q //= 2
    c = tp.rand_case(rng, B, Q, H, D, levels, P, lo=lo, hi=hi, dtype=np.float64 if f64 else np.float32)
    if kind == 4:  # a hot spot: most samples of one level in one cell
        hot = rng.uniform(0.2, 0.8, size=2)
        m = rng.uniform(size=c["loc"].shape[:-1]) < 0.8
        c["loc"][m] = (hot + rng.normal(0, 0.002, size=(int(m.sum()), 2))).astype(c["loc"].dtype)
    td = torch.float64 if f64 else torch.float32
    opts = {"value_path": int(rng.choice([0, 2, 3])), "place_path": int(rng.choice([0, 0, 1])),
            "q_round": int(rng.choice([0, 0, 0, max(1, Q // 3)])), "small_ns": int(rng.choice([0, 0, 1, 2, 3, 5])),
            # round 5: the LDS-served-level variants of the gather kernels (2: whenever a row fits) and the one-wave-per-unit
            # forward (2: every problem it can serve)
            "lds_levels": int(rng.choice([1, 1, 2, 2, 0])), "unit_fwd": int(rng.choice([1, 1, 0, 2])),
            # ... two planes per LDS-level workgroup (2: whenever H is even), block order (linear from N workgroups per plane)
            "lds_planes": int(rng.choice([0, 2, 2, 1])), "linear_slots": int(rng.choice([320, 1, 6, 40])), "touch": int(rng.choice([1, 0, 2])), "unit_waves": int(rng.choice([1, 2])),
            # round 6: passes over the batch (the workspace queries size for n passes; the backward follows the workspace)
            "ws_passes": int(rng.choice([1, 1, 2, 3])),
            # ... and padded value rows: elements behind every pixel's H * D channels (0: dense; values whose byte stride is not
            # a multiple of 16 take the dense-copy route of functional._value_rows — also a path worth fuzzing)
            "value_pad": int(rng.choice([0, 0, 4, 8, 32, 5]))}
    desc = dict(seed=seed, B=B, Q=Q, H=H, D=D, levels=levels, P=P, range=(lo, hi), pm=pm, ac=ac, f64=f64, **opts)
    return c, pm, ac, td, opts, desc, kind


OPTION_DEFAULTS = {"lds_levels": 1, "unit_fwd": 1, "lds_planes": 0, "linear_slots": 320, "touch": 1, "unit_waves": 1, "ws_passes": 1}


def run_case(c, pm, ac, td, opts):
    try:
        for k, v in opts.items():
            if k == "value_pad":  # not a library option: the layout of the tensor the test hands over (tests/test_gpu_parity.run_hip)
                os.environ["MSDA_TEST_VALUE_PAD"] = str(v)
            else:
                _lib.set_option(k, v)
        tp.check_against_oracle(msda_oracle, c, pm, ac, tp.FWD_TOL[td], tp.BWD_TOL[td])
    finally:
        os.environ.pop("MSDA_TEST_VALUE_PAD", None)
        for k in opts:
            if k != "value_pad":
                _lib.set_option(k, OPTION_DEFAULTS.get(k, 0))


msda_oracle.build()
if len(sys.argv) > 1 and sys.argv[1] == "--repro":  # fuzz_parity.py --repro SEED [TIMES] [k=v ...]: one case, repeated
    seed, times = int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 10
    c, pm, ac, td, opts, desc, _ = make_case(seed)
    extra = {}
    for kv in sys.argv[4:]:
        k, v = kv.split("=")
        extra[k] = int(v)
    opts.update({k: v for k, v in extra.items() if k in opts})
    pre = {k: v for k, v in extra.items() if k not in opts}
    for k, v in pre.items():
        _lib.set_option(k, v)
    bad = 0
    for i in range(times):
        try:
            run_case(c, pm, ac, td, opts)
        except AssertionError as e:
            bad += 1
            print("FAIL run", i, str(e).strip().splitlines()[0:5], flush=True)
    print(f"repro {json.dumps(desc)} with {opts} {pre}: {bad} of {times} runs failed")
    sys.exit(1 if bad else 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
n = fails = 0
seen = {}
seed = seed0
while time.time() - t0 < budget:
    c, pm, ac, td, opts, desc, kind = make_case(seed)
    try:
        run_case(c, pm, ac, td, opts)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("FAIL", json.dumps(desc), "::", str(e).strip().splitlines()[0:6], flush=True)
        if not isinstance(e, AssertionError):
            traceback.print_exc()
    n += 1
    seen[kind] = seen.get(kind, 0) + 1
    seed += 1
    if n % 50 == 0:
        print(f"... {n} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz: {n} cases (seeds {seed0}..{seed - 1}), {fails} failures, kinds {seen}, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
