"""dev tool (GPU box): bench.py's `padded_value_rows` leg alone — the full autograd step (forward, rand_like, backward) with
`img` dense / in padded rows, alternated in one process — under library options given as k=v arguments, e.g.
    python tools/padded_step_ab.py c2_q10k records_in_grads=0
(records_in_grads=0: the sample-gradient kernel runs FIRST in the backward instead of last)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from msda_triton_amd import _lib

wl = next((a for a in sys.argv[1:] if "=" not in a), "c2_q10k")
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
dev = torch.device("cuda", 0)
r = bench.padded_rows_leg(wl, dev)
print(wl, " ".join(a for a in sys.argv[1:] if "=" in a) or "(defaults)", json.dumps({k: r[k] for k in ("fwd_bwd_ms", "fwd_bwd_ms_rounds", "kernel_us")}))
