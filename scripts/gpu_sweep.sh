#!/bin/bash
# chunk-size sweep of the forward, then the full-size (100M-row) search
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
for c in 0 512 256 128 64 32; do
  timeout -k 10 200 python "$R/bench.py" --chunk $c --no-search --no-cpu --no-latency --steps 10 --warmup 3 > "$R/gpurun_out/sweep_chunk_$c.json" 2> "$R/gpurun_out/sweep_chunk_$c.err" || { echo "chunk $c failed"; tail -5 "$R/gpurun_out/sweep_chunk_$c.err"; exit 1; }
  python - "$R/gpurun_out/sweep_chunk_$c.json" <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print("chunk",r["config"]["batch_chunk"],"faces/s",r["value"],"ms",r["ms_per_step"],"sum_kernel_ms",r["sum_kernel_ms"])
PY
done
timeout -k 10 500 python "$R/bench.py" --no-cpu --no-latency --steps 5 --warmup 2 --gallery-rows 100000000 --search-steps 2 > "$R/gpurun_out/bench_100m.json" 2> "$R/gpurun_out/bench_100m.err" || { echo "100M failed"; tail -20 "$R/gpurun_out/bench_100m.err"; exit 1; }
python - "$R/gpurun_out/bench_100m.json" <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print(json.dumps(r["search"]))
PY
