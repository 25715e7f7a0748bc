#!/bin/bash
# dev tool: SAME-BOX A/B of builds of libmsda_hip.so, alternated (boxes differ by a few percent, which hides 1-3 %
# effects when the variants run in different gpurun calls).  Put the builds next to the library as
# msda_triton_amd/libmsda_hip_<name>.so (they travel with the snapshot; remove them afterwards), then on the GPU box:
#   [W=workload] bash tools/ab.sh base variant [variant2 ...]
# The shipped library is put back on ANY exit.
set -e
cd $GRAFT_REPO_ROOT
W=${W:-c2_q10k}
KEEP=$(mktemp /tmp/libmsda_hip_keep.XXXXXX)
cp msda_triton_amd/libmsda_hip.so $KEEP
trap 'cp $KEEP msda_triton_amd/libmsda_hip.so; rm -f $KEEP' EXIT
for n in "$@"; do test -f msda_triton_amd/libmsda_hip_$n.so || { echo "missing build $n"; exit 1; }; done
run() {
  cp msda_triton_amd/libmsda_hip_$1.so msda_triton_amd/libmsda_hip.so
  for i in 1 2 3; do timeout -k 10 200 python bench.py --workload $W --no-configs --no-do-bench --no-triton --no-cpu-baseline --no-strong-c5 --no-shard-compute 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %-10s fwd %.4f step %.4f' % ('$1', d['fwd_ms'], d['ms_per_step']), {k: v['avg_us'] for k, v in d['kernels'].items()})"; done
}
for rep in 1 2; do for n in "$@"; do run $n; done; done
