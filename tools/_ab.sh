#!/bin/bash
# same-box A/B of builds of libmsda_hip.so (dev)
cd $GRAFT_REPO_ROOT
run() {
  cp msda_triton_amd/libmsda_hip_$1.so msda_triton_amd/libmsda_hip.so
  for i in 1 2 3; do timeout -k 10 200 python bench.py --no-configs --no-do-bench --no-triton --no-cpu-baseline --no-strong-c5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $1 warm 10k fwd %.4f step %.4f' % (d['fwd_ms'], d['ms_per_step']), {k: v['avg_us'] for k, v in d['kernels'].items()})"; done
}
run hoist; run both; run hoist; run both
cp msda_triton_amd/libmsda_hip_both.so msda_triton_amd/libmsda_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixed.py -x -q 2>&1 | tail -2
