#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 kernel stats + HBM PMC passes of `python bench.py --workload W` into
# gpurun_out/<tag>_<W>_{stats,fetch,write}/.  Summaries go to profiles/ with tools/summarise_profiles.py <tag> <W>.
#   bash tools/collect_profiles.sh r02 c2_q10k [c1_readme ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r03}; shift
for W in "${@:-c2_q10k}"; do
  STEPS=20; [ "$W" = "c5_stress" ] && STEPS=4
  # overlap=0: every kernel runs alone, as in bench.py's per-kernel HIP-event timing (by default the sample-gradient
  # kernel overlaps the grad_value pipeline on large / c4-sized problems, which stretches both kernels' own durations)
  B="python bench.py --workload $W --steps $STEPS --warmup 3 --no-cpu-baseline --no-strong-c5 --no-shard-compute --no-configs --no-do-bench --no-triton --opt overlap=0"
  rm -rf gpurun_out/${TAG}_${W}_stats gpurun_out/${TAG}_${W}_fetch gpurun_out/${TAG}_${W}_write
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_${W}_stats -- $B > gpurun_out/${TAG}_${W}_stats.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_${W}_fetch -- python bench.py --workload $W --steps 3 --warmup 2 --spin-up-ms 0 --no-cpu-baseline --no-strong-c5 --no-shard-compute --no-configs --no-do-bench --no-triton > gpurun_out/${TAG}_${W}_fetch.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_${W}_write -- python bench.py --workload $W --steps 3 --warmup 2 --spin-up-ms 0 --no-cpu-baseline --no-strong-c5 --no-shard-compute --no-configs --no-do-bench --no-triton > gpurun_out/${TAG}_${W}_write.log 2>&1
  echo "$W done: $(tail -c 300 gpurun_out/${TAG}_${W}_stats.log | grep -o '"fwd_bwd_ms": [0-9.]*')"
done
