#!/bin/bash
# dev tool: per-kernel times of the backward value path under a debug mask
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in "$@"; do
  rm -rf gpurun_out/prof_dbg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dbg -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt debug=$dbg > gpurun_out/prof_dbg.log 2>&1
  echo "== debug=$dbg"
  python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_dbg/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'msda' in r['Name']:
            print(f"{r['Name'].split('msda::')[1].split('(')[0]:55s} {float(r['AverageNs'])/1000:9.1f} us  x{r['Calls']}")
PY
done
