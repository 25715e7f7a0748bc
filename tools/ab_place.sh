#!/bin/bash
# dev tool: per-kernel times of the sorted pipeline under the place-pass variants (place_path 1 plane-major, 2 level-major
# 1024 threads, 3 level-major 256 threads), alternating on one box:  bash tools/ab_place.sh dec_coco [extra --opt ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-dec_coco}; shift
for rep in 1 2; do
  for pp in 1 2 3; do
    rm -rf gpurun_out/prof_dbg
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dbg -- python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-strong-c5 --no-shard-compute --no-configs --no-do-bench --no-triton --opt place_path=$pp "$@" > gpurun_out/prof_dbg.log 2>&1
    echo "== $W place_path=$pp: $(grep -o '"fwd_bwd_ms": [0-9.]*' gpurun_out/prof_dbg.log | head -1) $(bash tools/kstats.sh gpurun_out/prof_dbg | grep -i 'place\|pass' | awk '{printf "%s %s us | ", $1, $(NF-2)}')"
  done
done
