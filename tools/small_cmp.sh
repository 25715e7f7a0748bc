#!/bin/bash
# dev tool: single-launch grad_value kernel time + step time for the small workloads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in c4_gdino_dec c1_readme c2_q1k; do
  rm -rf gpurun_out/prof_dbg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dbg -- python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-strong-c5 --no-configs --no-do-bench --no-triton "$@" > gpurun_out/prof_dbg.log 2>&1
  echo "== $w: $(grep -o '"fwd_bwd_ms": [0-9.]*' gpurun_out/prof_dbg.log | head -1) $(bash tools/kstats.sh gpurun_out/prof_dbg | grep value_small | awk '{print $(NF-2), $(NF-1)}')"
done
