#!/bin/bash
# model tests, then a short forward-only bench with the per-kernel table.  Usage: gpu_quick.sh <tag> [bench args]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash "$R/scripts/gpu_tests.sh" tests/test_gpu_model.py || exit 1
timeout -k 10 300 python "$R/bench.py" --no-search --no-cpu --no-latency --steps 10 --warmup 3 "$@" > "$R/gpurun_out/bench_$tag.json" 2> "$R/gpurun_out/bench_$tag.err" || { tail -5 "$R/gpurun_out/bench_$tag.err"; exit 1; }
python - "$R/gpurun_out/bench_$tag.json" <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print("faces/s", r["value"], "ms/step", r["ms_per_step"], "sum_kernel_ms", r["sum_kernel_ms"])
for k,v in sorted(r["kernels"].items(), key=lambda kv:-kv[1]["ms"]): print("  %-55s n=%2d %8.4f ms %8.1f GB/s %7.2f TF" % (k, v["launches"], v["ms"], v["GBps"], v["TFLOPs"]))
PY
