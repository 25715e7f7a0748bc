#!/bin/bash
# dev tool: per-kernel times (rocprofv3 kernel trace) of bench.py under option settings, e.g. "cell_slices=32"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for opt in "$@"; do
  rm -rf gpurun_out/prof_opt
  args=""
  for kv in ${opt//,/ }; do args="$args --opt $kv"; done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_opt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c5 $args > gpurun_out/prof_opt.log 2>&1
  echo "== $opt"
  python - <<'PY'
import csv, glob, os, re
f = max(glob.glob('gpurun_out/prof_opt/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'msda' not in n:
        continue
    m = re.search(r'msda_\w+?_kernel', n)
    tail = n.split(m.group(0), 1)[1].split('(')[0][:40] if m else ''
    print(f"{(m.group(0) if m else n)[:40] + tail:70s} {float(r['AverageNs'])/1000:9.1f} us  x{r['Calls']}")
PY
done
