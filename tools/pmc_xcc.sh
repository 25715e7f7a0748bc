#!/bin/bash
# dev tool: per-XCC (and per-instance) values of a few texture-path / L2 counters for the forward kernel — does the XCD that
# serves the slow head (HISTORY §4 item 5) differ in tag conflicts, pending stalls or L2 channel load?
#   [OPTS="--opt lds_levels=0"] [INSTANCES=1] bash tools/pmc_xcc.sh "TCP_TAGRAM0_REQ TCP_TAGRAM1_REQ TCP_TAGRAM2_REQ TCP_TAGRAM3_REQ" [kernel substring]
#   PROG="python tools/row_stride_ab.py --pads 128 --rounds 1 --reps 3 --no-spin": profile that program instead of bench.py
# (the TA_BUFFER_* counters hang the profiler on this image: do not ask for them)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C="$1"; K=${2:-msda_fwd_kernel}
rm -rf gpurun_out/pmc_xcc
if [ -n "$PROG" ]; then
  timeout -k 10 120 rocprofv3 --pmc $C --output-format json -d gpurun_out/pmc_xcc -- $PROG > gpurun_out/pmc_xcc.log 2>&1
else
  timeout -k 10 120 rocprofv3 --pmc $C --output-format json -d gpurun_out/pmc_xcc -- python bench.py --workload ${W:-c2_q10k} --steps 2 --warmup 1 --spin-up-ms 0 --no-cpu-baseline --no-strong-c5 --no-shard-compute --no-configs --no-do-bench --no-triton $OPTS > gpurun_out/pmc_xcc.log 2>&1
fi
python - "$K" <<'PY'
import collections, glob, json, os, sys
want = sys.argv[1]
f = max(glob.glob('gpurun_out/pmc_xcc/*/*results.json'), key=os.path.getmtime)
d = json.load(open(f))["rocprofiler-sdk-tool"][0]
kern = {k["kernel_id"]: k.get("formatted_kernel_name", "") for k in d["kernel_symbols"]}
ctr = {c["id"]["handle"]: c for c in d["counters"]}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = 0
for rec in d["callback_records"]["counter_collection"]:
    if want not in kern.get(rec["dispatch_data"]["dispatch_info"]["kernel_id"], ""):
        continue
    n += 1
    per = collections.defaultdict(list)
    for r in rec["records"]:
        per[r["counter_id"]["handle"]].append(r["value"])
    for cid, vals in per.items():  # (the records of a counter come in the order of its `instances`)
        for i, v in enumerate(vals):
            agg[cid][i] += v
print("dispatches of", want, ":", n)
for cid, vals in agg.items():
    c = ctr[cid]
    by_x, by_xi = collections.defaultdict(float), collections.defaultdict(float)
    for i, v in vals.items():
        dm = {x["dimension_name"]: x["index"] for x in c["instances"][i]["dimensions"]}
        by_x[dm.get("DIMENSION_XCC", -1)] += v / n
        by_xi[(dm.get("DIMENSION_XCC", -1), dm.get("DIMENSION_INSTANCE", -1))] += v / n
    print("%-36s per XCC: %s" % (c["name"], {x: int(v) for x, v in sorted(by_x.items())}))
    if os.environ.get("INSTANCES"):
        for x in sorted(by_x):
            print("    XCC %d by instance: %s" % (x, [int(v) for (xx, i), v in sorted(by_xi.items()) if xx == x]))
PY
