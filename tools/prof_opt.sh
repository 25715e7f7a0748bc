#!/bin/bash
# dev tool: per-kernel times (rocprofv3 kernel trace) of bench.py under option settings, e.g. "cell_slices=32"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for opt in "$@"; do
  rm -rf gpurun_out/prof_opt
  args=""
  for kv in ${opt//,/ }; do args="$args --opt $kv"; done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_opt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline $args > gpurun_out/prof_opt.log 2>&1
  echo "== $opt"
  python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_opt/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'msda' in r['Name']:
            print(f"{r['Name'].split('msda::')[1].split('(')[0]:55s} {float(r['AverageNs'])/1000:9.1f} us  x{r['Calls']}")
PY
done
