#!/bin/bash
# dev tool: a few SQ counter passes of bench.py (options as args "k=v,k=v"), printed per msda kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args=""
for kv in ${1//,/ }; do args="$args --opt $kv"; done
SETS=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN"
  "GRBM_GUI_ACTIVE TA_BUSY_avr"
)
n=0
for g in "${SETS[@]}"; do
  rm -rf gpurun_out/pq_$n
  timeout -k 10 200 rocprofv3 --pmc $g --output-format csv -d gpurun_out/pq_$n -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-strong-c5 --no-shard-compute --no-configs --no-do-bench --no-triton $args > gpurun_out/pq_$n.log 2>&1 || echo "set $n failed"
  n=$((n+1))
done
python3 - <<'PY'
import csv, glob, collections, re, os
FILTER = os.environ.get('KFILTER', '')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pq_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'msda' not in k or (FILTER and FILTER not in k): continue
        k = re.sub(r'^void msda::', '', k).split('(')[0]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print('==', k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f'   {c:28s} {sum(v)/len(v):14.0f}  (n={len(v)})')
PY
