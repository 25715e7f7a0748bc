#!/bin/bash
# dev tool: every kernel of the module's training step (tools/module_step.py) with its share, as a CSV on stdout:
#   bash tools/prof_module_step.sh [module_step.py arguments] > profiles/rNN_module_step_kernels.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_mstep
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mstep -- python tools/module_step.py "$@" 38 > gpurun_out/prof_mstep.log 2>&1
python - <<'PY'
import csv, glob, re
rows = []
for f in glob.glob('gpurun_out/prof_mstep/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Name']
        nm = re.sub(r'^void ', '', n)
        nm = nm.split('msda::')[1].split('(')[0] if 'msda::' in nm else nm[:110]
        rows.append((float(r['TotalDurationNs']), int(r['Calls']), float(r['AverageNs']), nm))
rows.sort(reverse=True)
steps = float(max([c for _, c, _, nm in rows if 'msda_fwd_kernel' in nm or 'msda_fwd_unit_kernel' in nm] or [46]))  # one forward per step
tot = sum(r[0] for r in rows)
print("kernel,calls_per_step,avg_us,us_per_step,share")
for t, c, a, nm in rows:
    print('"%s",%.2f,%.2f,%.2f,%.4f' % (nm.replace('"', "'"), c / steps, a / 1e3, t / 1e3 / steps, t / tot))
print('"TOTAL device time per step",,,%.2f,1.0' % (tot / 1e3 / steps))
PY
