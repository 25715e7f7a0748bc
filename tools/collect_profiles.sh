#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 kernel stats + HBM PMC passes of `python bench.py` into gpurun_out/.
# Copy the summaries into profiles/ afterwards with tools/summarise_profiles.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rm -rf gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- $B > gpurun_out/${TAG}_stats.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_write.log 2>&1
ls gpurun_out/${TAG}_*/*/ | head
