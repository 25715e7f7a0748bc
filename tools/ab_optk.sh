#!/bin/bash
# dev tool: same-box alternating A/B of option strings ("k=v,k=v" or "-"), per-KERNEL times from the library's own event
# pairs (bench.py single_kernels):  bash tools/ab_optk.sh c2_q10k - cell_slices=16
cd $GRAFT_REPO_ROOT
W=$1; shift
for rep in 1 2; do
  for o in "$@"; do
    args=""; [ "$o" != "-" ] && for kv in ${o//,/ }; do args="$args --opt $kv"; done
    for i in 1 2; do timeout -k 10 200 python bench.py --workload $W --no-configs --no-do-bench --no-triton --no-cpu-baseline --no-strong-c5 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %-16s step %.4f' % ('$o', d['ms_per_step']), {k.replace('msda_','').replace('_kernel',''): round(v['avg_us'],1) for k, v in d['single_kernels'].items()})"; done
  done
done
