#!/bin/bash
# dev tool: per-kernel times of a workload under option strings, alternating on one box:  bash tools/ab_dbg.sh c3_ddetr_enc debug=0 debug=4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=$1; shift
STEPS=${STEPS:-20}
for rep in 1 2; do
  for o in "$@"; do
    args=""; [ "$o" != "-" ] && for kv in ${o//,/ }; do args="$args --opt $kv"; done
    rm -rf gpurun_out/prof_dbg
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dbg -- python bench.py --workload $W --steps $STEPS --warmup 3 --no-cpu-baseline --no-strong-c5 --no-shard-compute --no-configs --no-do-bench --no-triton $args > gpurun_out/prof_dbg.log 2>&1
    echo "== $W $o: $(bash tools/kstats.sh gpurun_out/prof_dbg | grep -v 'fwd_kernel\|bwd_sample' | awk '{printf "%s %s | ", substr($1,6,14), $(NF-2)}')"
  done
done
