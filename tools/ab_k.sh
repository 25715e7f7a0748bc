#!/bin/bash
# dev tool: same-box A/B of BUILDS (msda_triton_amd/libmsda_hip_<name>.so), per-KERNEL times from the library's own
# event pairs (bench.py single_kernels), alternated twice:  [W=workload] [REPS=2] bash tools/ab_k.sh base variant [variant2 ...]
# The shipped library is put back on ANY exit (interrupt, timeout, missing variant).
set -e
cd $GRAFT_REPO_ROOT
W=${W:-c2_q10k}
REPS=${REPS:-2}
KEEP=$(mktemp /tmp/libmsda_hip_keep.XXXXXX)
cp msda_triton_amd/libmsda_hip.so $KEEP
trap 'cp $KEEP msda_triton_amd/libmsda_hip.so; rm -f $KEEP' EXIT
for n in "$@"; do test -f msda_triton_amd/libmsda_hip_$n.so || { echo "missing build $n"; exit 1; }; done
run() {
  cp msda_triton_amd/libmsda_hip_$1.so msda_triton_amd/libmsda_hip.so
  for i in $(seq $REPS); do timeout -k 10 200 python bench.py --workload $W --no-configs --no-do-bench --no-triton --no-cpu-baseline --no-strong-c5 --no-shard-compute 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %-8s fwd %.4f step %.4f' % ('$1', d['fwd_ms'], d['ms_per_step']), {k.replace('msda_','').replace('_kernel',''): round(v['avg_us'],1) for k, v in d['single_kernels'].items()})"; done
}
for rep in 1 2; do for n in "$@"; do run $n; done; done
