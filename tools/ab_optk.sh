#!/bin/bash
# dev tool: same-box alternating A/B of option strings ("k=v,k=v" or "-"), per-KERNEL times from the library's own event
# pairs (bench.py single_kernels):  [LIB=<build name>] [REPS=2] bash tools/ab_optk.sh c2_q10k - cell_slices=16
# LIB: run msda_triton_amd/libmsda_hip_<name>.so instead of the shipped library (put back on any exit).
set -e
cd $GRAFT_REPO_ROOT
W=$1; shift
REPS=${REPS:-2}
if [ -n "$LIB" ]; then
  KEEP=$(mktemp /tmp/libmsda_hip_keep.XXXXXX)
  cp msda_triton_amd/libmsda_hip.so $KEEP
  trap 'cp $KEEP msda_triton_amd/libmsda_hip.so; rm -f $KEEP' EXIT
  cp msda_triton_amd/libmsda_hip_$LIB.so msda_triton_amd/libmsda_hip.so
fi
for rep in 1 2; do
  for o in "$@"; do
    args=""; [ "$o" != "-" ] && for kv in ${o//,/ }; do args="$args --opt $kv"; done
    for i in $(seq $REPS); do timeout -k 10 200 python bench.py --workload $W --no-configs --no-do-bench --no-triton --no-cpu-baseline --no-strong-c5 --no-shard-compute $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %-16s fwd %.4f step %.4f' % ('$o', d['fwd_ms'], d['ms_per_step']), {k.replace('msda_','').replace('_kernel',''): round(v['avg_us'],1) for k, v in d['single_kernels'].items() if 'fwd' in k or 'bwd_sample' in k or '$ALLK'})"; done
  done
done
