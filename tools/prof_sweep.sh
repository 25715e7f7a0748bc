#!/bin/bash
# dev tool: per-kernel times (rocprofv3 kernel trace) of tools/sweep.py runs; each arg = "workload[:k=v...] opt=val ..." (quote it)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  rm -rf gpurun_out/prof_sw
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sw -- python tools/sweep.py $spec > gpurun_out/prof_sw.log 2>&1
  echo "== $spec"
  tail -1 gpurun_out/prof_sw.log | cut -c1-200
  python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_sw/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'msda' in r['Name']:
            print(f"{r['Name'].split('msda::')[1].split('(')[0]:55s} {float(r['AverageNs'])/1000:9.1f} us  x{r['Calls']}")
PY
done
