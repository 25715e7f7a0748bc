#!/bin/bash
# dev tool: per-kernel times of the binned grad_value path under debug (ablation) masks
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in "$@"; do
  rm -rf gpurun_out/prof_dbg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dbg -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strong-c5 --opt value_path=4 --opt overlap=0 --opt debug=$dbg > gpurun_out/prof_dbg.log 2>&1
  echo "== debug=$dbg"
  bash tools/kstats.sh gpurun_out/prof_dbg | grep -E "bin_pass|tile_"
done
