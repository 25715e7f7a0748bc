#!/bin/bash
# dev tool: same-box alternating A/B of two option strings ("k=v,k=v" or "-" for defaults):  bash tools/ab_opt.sh - overlap=0 [workload]
cd $GRAFT_REPO_ROOT
W=${3:-c2_q10k}
run() {
  args=""; [ "$1" != "-" ] && for kv in ${1//,/ }; do args="$args --opt $kv"; done
  for i in 1 2 3; do timeout -k 10 200 python bench.py --workload $W --no-configs --no-do-bench --no-triton --no-cpu-baseline --no-strong-c5 --no-shard-compute $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %-14s fwd %.4f step %.4f' % ('$1', d['fwd_ms'], d['ms_per_step']))"; done
}
run "$1"; run "$2"; run "$1"; run "$2"
