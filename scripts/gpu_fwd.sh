#!/bin/bash
# short forward-only bench with the per-kernel table (no tests).  Usage: gpu_fwd.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
timeout -k 10 300 python "$R/bench.py" --no-search --no-cpu --no-latency --steps 10 --warmup 3 "$@" > "$R/gpurun_out/bench_fwd.json" 2> "$R/gpurun_out/bench_fwd.err" || { tail -5 "$R/gpurun_out/bench_fwd.err"; exit 1; }
python - "$R/gpurun_out/bench_fwd.json" <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print("faces/s", r["value"], "ms/step", r["ms_per_step"], "sum_kernel_ms", r["sum_kernel_ms"])
for k,v in sorted(r["kernels"].items(), key=lambda kv:-kv[1]["ms"]): print("  %-55s n=%2d %8.4f ms" % (k, v["launches"], v["ms"]))
PY
