#!/bin/bash
# dev tool: per-kernel times (rocprofv3 kernel trace) of an arbitrary python script: bash tools/prof_any.sh script.py [args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_any
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_any -- python "$@" > gpurun_out/prof_any.log 2>&1
python - <<'PY'
import csv, glob, os, re
f = max(glob.glob('gpurun_out/prof_any/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'msda' not in n:
        continue
    m = re.search(r'msda_\w+?_kernel', n)
    tail = n.split(m.group(0), 1)[1].split('(')[0][:40] if m else ''
    print(f"{(m.group(0) if m else n)[:40] + tail:70s} {float(r['AverageNs'])/1000:9.1f} us  x{r['Calls']}")
PY
