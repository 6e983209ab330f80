#!/bin/bash
# bench + rocprofv3 kernel-trace summary of a short run.  Usage: gpu_bench_prof.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
timeout -k 10 600 python "$R/bench.py" "$@" > "$R/gpurun_out/bench_$tag.json" 2> "$R/gpurun_out/bench_$tag.err" || { echo "bench failed"; tail -20 "$R/gpurun_out/bench_$tag.err"; exit 1; }
cat "$R/gpurun_out/bench_$tag.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$tag" -- python3 "$R/bench.py" --steps 100 --warmup 20 --no-search --no-cpu --no-latency > "$R/gpurun_out/prof_$tag.log" 2>&1 || { echo "rocprof failed"; tail -20 "$R/gpurun_out/prof_$tag.log"; exit 1; }
find "$R/gpurun_out/prof_$tag" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$R/gpurun_out/kernel_stats_$tag.csv"
head -25 "$R/gpurun_out/kernel_stats_$tag.csv"
