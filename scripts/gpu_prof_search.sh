#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
rows=${1:-1000000}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_search" -- python3 "$R/bench.py" --search-only --gallery-rows $rows --search-steps 3 > "$R/gpurun_out/prof_search.log" 2>&1 || { tail -20 "$R/gpurun_out/prof_search.log"; exit 1; }
tail -1 "$R/gpurun_out/prof_search.log"
f=$(find "$R/gpurun_out/prof_search" -name "*kernel_stats.csv" | head -1)
grep -E "fra|Name" "$f" | cut -c1-200
