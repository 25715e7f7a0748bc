#!/bin/bash
# dev tool: per-kernel times of the binned grad_value path for a list of option strings ("k=v,k=v")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${WORKLOAD:-c2_q10k}
for o in "$@"; do
  args=""
  for kv in ${o//,/ }; do args="$args --opt $kv"; done
  rm -rf gpurun_out/prof_dbg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dbg -- python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-strong-c5 --opt value_path=4 --opt overlap=0 $args > gpurun_out/prof_dbg.log 2>&1
  echo "== $o"
  bash tools/kstats.sh gpurun_out/prof_dbg | grep -E "bin_pass|tile_"
done
