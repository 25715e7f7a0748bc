#!/bin/bash
# dev tool: per-kernel time of the single-launch grad_value kernel under option strings, alternating on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for W in c4_gdino_dec c1_readme c2_q1k; do
  for rep in 1 2; do
    for o in debug=0 debug=2 debug=1; do
      rm -rf gpurun_out/prof_dbg
      timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dbg -- python bench.py --workload $W --steps 30 --warmup 3 --no-cpu-baseline --no-strong-c5 --no-configs --no-do-bench --no-triton --opt $o > gpurun_out/prof_dbg.log 2>&1
      echo "== $W $o: $(grep -o '"fwd_bwd_ms": [0-9.]*' gpurun_out/prof_dbg.log | head -1) $(bash tools/kstats.sh gpurun_out/prof_dbg | grep small)"
    done
  done
done
