#!/bin/bash
# dev tool: per-dispatch durations of one backward (kernel trace), under a debug mask
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --opt debug=$1 > gpurun_out/prof_trace.log 2>&1
python - <<'PY'
import csv,glob
rows=[]
for f in glob.glob('gpurun_out/prof_trace/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']))
rows.sort()
last=rows[-40:]
t0=last[0][0]
for s,e,n in last:
    nm=n.split('msda::')[1].split('(')[0] if 'msda::' in n else n[:40]
    print(f"{(s-t0)/1000:9.1f} us  +{(e-s)/1000:8.1f} us  {nm[:60]}")
PY
