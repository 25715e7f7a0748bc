#!/bin/bash
# dev tool: start/end timeline (us, relative) of the msda kernels of the last two bench steps under option strings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${WORKLOAD:-c2_q10k}
for o in "$@"; do
  args=""
  for kv in ${o//,/ }; do args="$args --opt $kv"; done
  rm -rf gpurun_out/prof_tl
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python bench.py --workload $W --steps 4 --warmup 2 --no-cpu-baseline --no-strong-c5 --no-shard-compute --no-configs --no-do-bench --no-triton $args > gpurun_out/prof_tl.log 2>&1
  echo "== $W $o: $(grep -o '"fwd_bwd_ms": [0-9.]*' gpurun_out/prof_tl.log | head -1)"
  python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_tl/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the timed (un-instrumented) steps: take a window in the middle of the run
ms=[r for r in rows if 'msda' in r['Kernel_Name'] or 'at::native' in r['Kernel_Name']]
fw=[i for i,r in enumerate(ms) if 'msda_fwd_kernel' in r['Kernel_Name']]
i0=fw[3]; i1=fw[5] if len(fw)>5 else len(ms)
t0=int(ms[i0]['Start_Timestamp'])
for r in ms[i0:i1]:
    n=r['Kernel_Name'].replace('void msda::','').split('(')[0][:44]
    s=(int(r['Start_Timestamp'])-t0)/1000; e=(int(r['End_Timestamp'])-t0)/1000
    print(f"  {n:46s} {s:8.1f} -> {e:8.1f}  ({e-s:6.1f} us) q{r['Queue_Id']}")
PY
done
