#!/bin/bash
# dev tool: print the msda kernels of the newest rocprofv3 kernel_stats.csv under a directory
f=$(ls -t $1/*/*kernel_stats.csv | head -1)
grep msda $f | awk -F'","|",|,' '{n=$1; gsub(/"/,"",n); sub(/^_ZN4msda[0-9]+/,"",n); printf "%-70s %9.1f us x%s\n", substr(n,1,70), $4/1000, $2}'
