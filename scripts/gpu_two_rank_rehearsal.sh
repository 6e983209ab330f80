#!/bin/bash
# The multi-rank path of bench.py rehearsed on ONE GPU: two ranks on cuda:0, gloo in place of RCCL (which refuses two
# ranks on one device).  Prints rank 0's line; its search.result_digest must equal the one-rank run's.
mkdir -p gpurun_out
export FRA_BENCH_ONE_GPU=1 FRA_BENCH_BACKEND=gloo FRA_BENCH_WATCHDOG=240
timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 3 --warmup 1 --gallery-rows 3000000 --queries 2000 --search-steps 2 --no-cpu --no-latency > gpurun_out/two_rank.json 2> gpurun_out/two_rank.err || { tail -20 gpurun_out/two_rank.err; exit 1; }
unset FRA_BENCH_ONE_GPU FRA_BENCH_BACKEND
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --gallery-rows 3000000 --queries 2000 --search-steps 2 --no-cpu --no-latency > gpurun_out/one_rank.json 2> gpurun_out/one_rank.err || { tail -20 gpurun_out/one_rank.err; exit 1; }
python - <<'PY'
import json
a = json.loads(open("gpurun_out/two_rank.json").read().strip().splitlines()[-1])
b = json.loads(open("gpurun_out/one_rank.json").read().strip().splitlines()[-1])
print("two ranks:", a["n_gpus"], a["value"], a["search"]["result_digest"], a["search"]["shard"], a["search"]["merge"])
print("one rank :", b["n_gpus"], b["value"], b["search"]["result_digest"], b["search"]["shard"])
assert a["search"]["result_digest"] == b["search"]["result_digest"], "digests differ"
print("digests equal")
PY
