#!/bin/bash
# dev tool: per-kernel times (rocprofv3 kernel trace) of bench.py on the given workloads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in "$@"; do
  rm -rf gpurun_out/prof_wl
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wl -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $wl --opt value_path=2 > gpurun_out/prof_wl.log 2>&1
  echo "== $wl"
  python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_wl/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'msda' in r['Name']:
            print(f"{r['Name'].split('msda::')[1].split('(')[0]:55s} {float(r['AverageNs'])/1000:9.1f} us  x{r['Calls']}")
PY
done
