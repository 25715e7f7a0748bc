#!/bin/bash
# dev tool (GPU box): run the commands given on stdin one after the other, each under its own `timeout -k 10 <seconds>`
# (first word of the line), output to gpurun_out/<tag>.log (second word); STOP at the first command that was killed at its
# limit or died on a signal — after a hung GPU step nothing else is started in the same gpurun call.
#   bash tools/steps.sh <<'EOS'
#   300 nccl_tests python -m pytest tests/test_gpu_nccl.py -x -q
#   EOS
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
while read -r limit tag cmd; do
  [ -z "$limit" ] && continue
  echo "== $tag: $cmd"
  timeout -k 10 $limit bash -c "$cmd" > gpurun_out/$tag.log 2>&1 < /dev/null
  rc=$?
  echo "   rc=$rc"; tail -n 4 gpurun_out/$tag.log | cut -c1-300
  if [ $rc -ge 124 ]; then echo "step $tag was killed (rc=$rc): stopping"; exit $rc; fi
done
exit 0
