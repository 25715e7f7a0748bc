#!/usr/bin/env python3
"""Dev tool: one-line digest of the last bench.py JSON line in gpurun_out/bench.log."""
import json
import sys
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench.log"
d = json.loads(open(path).read().strip().splitlines()[-1])
print("fwd_ms", round(d["fwd_ms"], 4), "fwd_bwd_ms", round(d["fwd_bwd_ms"], 4),
      {k: v["avg_us"] for k, v in d["kernels"].items()})
