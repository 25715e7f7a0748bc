#!/bin/bash
# dev tool: same-box A/B of two BUILDS (msda_triton_amd/libmsda_hip_<name>.so), per-KERNEL times from the library's own
# event pairs (bench.py single_kernels):  bash tools/ab_k.sh base variant [workload]
cd $GRAFT_REPO_ROOT
W=${3:-c2_q10k}
cp msda_triton_amd/libmsda_hip.so /tmp/libmsda_hip_keep.so
run() {
  cp msda_triton_amd/libmsda_hip_$1.so msda_triton_amd/libmsda_hip.so
  for i in 1 2; do timeout -k 10 200 python bench.py --workload $W --no-configs --no-do-bench --no-triton --no-cpu-baseline --no-strong-c5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %-8s step %.4f' % ('$1', d['ms_per_step']), {k.replace('msda_','').replace('_kernel',''): round(v['avg_us'],1) for k, v in d['single_kernels'].items()})"; done
}
run $1; run $2; run $1; run $2
cp /tmp/libmsda_hip_keep.so msda_triton_amd/libmsda_hip.so
