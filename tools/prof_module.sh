#!/bin/bash
# dev tool: per-kernel times (rocprofv3 kernel trace) of tools/time_module.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_mod
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mod -- python tools/time_module.py > gpurun_out/prof_mod.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_mod/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if float(r['TotalDurationNs'])>2e6:
            nm = n.split('msda::')[1].split('(')[0] if 'msda::' in n else n[:70]
            print(f"{nm:75s} {float(r['AverageNs'])/1000:9.1f} us  x{r['Calls']}")
PY
