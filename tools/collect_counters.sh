#!/bin/bash
# Run ON THE GPU BOX (gpurun): pipeline-utilisation counters of `python bench.py`, one rocprofv3 --pmc pass per
# counter group (no tracing options alongside), into gpurun_out/<tag>_pmc_<n>/.
# Summarise afterwards with tools/summarise_counters.py.   WL=<workload> and EXTRA="--opt k=v ..." select another
# workload / library options (e.g. WL=c3_ddetr_enc EXTRA="--opt lds_levels=0" bash tools/collect_counters.sh r05c3plain).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
W=${WL:-c2_q10k}
SETS=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY"
  "TA_BUSY_avr GRBM_GUI_ACTIVE"
  "VALUBusy"
  "MemUnitStalled"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
)
n=0
for g in "${SETS[@]}"; do
  rm -rf gpurun_out/${TAG}_pmc_$n
  timeout -k 10 300 rocprofv3 --pmc $g --output-format csv -d gpurun_out/${TAG}_pmc_$n -- python bench.py --workload $W $EXTRA --steps 3 --warmup 2 --no-cpu-baseline --no-strong-c5 --no-shard-compute --no-configs --no-do-bench --no-triton > gpurun_out/${TAG}_pmc_$n.log 2>&1 || echo "group $n ($g) failed: see gpurun_out/${TAG}_pmc_$n.log"
  n=$((n+1))
done
ls gpurun_out/${TAG}_pmc_*/*/ 2>/dev/null | head -40
