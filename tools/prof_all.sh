#!/bin/bash
# dev tool: per-kernel times (all msda kernels, forward included) of a workload under option strings ("k=v,k=v" or "-"):
#   WORKLOAD=c3_ddetr_enc bash tools/prof_all.sh lds_levels=0 -
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${WORKLOAD:-c2_q10k}
STEPS=${STEPS:-6}
for o in "$@"; do
  args=""
  [ "$o" != "-" ] && for kv in ${o//,/ }; do args="$args --opt $kv"; done
  rm -rf gpurun_out/prof_all
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_all -- python bench.py --workload $W --steps $STEPS --warmup 2 --no-cpu-baseline --no-strong-c5 --no-shard-compute --no-configs --no-do-bench --no-triton $args > gpurun_out/prof_all.log 2>&1
  echo "== $W $o: $(grep -o '"fwd_bwd_ms": [0-9.]*' gpurun_out/prof_all.log | head -1)"
  bash tools/kstats.sh gpurun_out/prof_all
done
