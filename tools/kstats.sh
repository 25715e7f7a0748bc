#!/bin/bash
# dev tool: print the msda kernels of the newest rocprofv3 kernel_stats.csv under a directory
f=$(ls -t $1/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'msda' in r['Name']:
        n = re.sub(r'^_ZN4msda\d+', '', r['Name'].replace('void msda::', '')).split('(')[0]
        print(f"{n[:70]:70s} {float(r['AverageNs'])/1000:9.1f} us x{r['Calls']}")
PY
