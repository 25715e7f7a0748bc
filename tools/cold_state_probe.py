"""dev tool: what flips the cold-cache forward between its two readings (22-26 us / 17-18 us at Q = 900)?  One process,
the sweep's do_bench (no spin-up), a sequence of events between measurements:
    python tools/cold_state_probe.py [Q]"""
import importlib.util
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msda_triton_amd import _lib, multiscale_deformable_attention  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("sweep", os.path.join(here, "..", "scripts", "benchmark_sweep.py"))
sweep = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sweep)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 900


def measure(tag, inputs):
    img, shapes, pts, att = inputs

    def fwd():
        with torch.no_grad():
            multiscale_deformable_attention(img, shapes, pts, att, "border", True)

    v = [sweep.do_bench(fwd, warmup_ms=30.0, rep_ms=200.0, spin_ms=0)[0] * 1e3 for _ in range(2)]
    print("%-58s %s us" % (tag, "/".join("%.2f" % x for x in v)), flush=True)


inp = sweep.make_inputs(N, False)
measure("fresh inputs", inp)
measure("again", inp)
time.sleep(3.0)
measure("after 3 s of idle", inp)
inp[0].sum().item()
measure("after img.sum()", inp)
inp = sweep.make_inputs(N, False)
measure("fresh inputs (2)", inp)
_lib.set_option("xcd_map", 0)
with torch.no_grad():
    multiscale_deformable_attention(*inp, "border", True)
torch.cuda.synchronize()
_lib.set_option("xcd_map", 1)
measure("after ONE call in linear block order", inp)
inp = sweep.make_inputs(N, False)
measure("fresh inputs (3)", inp)
big = torch.empty(1 << 30, dtype=torch.int8, device="cuda")
big.zero_()
torch.cuda.synchronize()
del big
measure("after zeroing 1 GiB", inp)
inp = tuple(t.clone() for t in inp)
measure("clones of the inputs", inp)
inp = sweep.make_inputs(N, False)
measure("fresh inputs (4)", inp)
for _ in range(3):
    measure("... again", inp)
