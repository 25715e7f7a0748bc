cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in 0 1; do
rm -rf gpurun_out/pl2_$o
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/pl2_$o -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-strong-c5 --no-shard-compute --no-configs --no-do-bench --no-triton --opt lds_levels=$o > gpurun_out/pl2_$o.log 2>&1
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pl2_$o/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'msda_fwd' not in k and 'bwd_sample' not in k: continue
        k = re.sub(r'^void msda::', '', k).split('<')[0]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print('lds_levels=$o', k, {c: int(sum(v)/len(v)) for c, v in sorted(acc[k].items())})
PY
done
