#!/bin/bash
# hardware counters for the forward's kernels.  Usage: gpu_pmc.sh <tag> "<counters>" [bench args]
tag=$1; ctrs=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_$tag" -- python3 "$R/bench.py" ${PMC_BENCH_ARGS:---steps 2 --warmup 1 --no-search --no-cpu --no-latency} "$@" > "$R/gpurun_out/pmc_$tag.log" 2>&1 || { echo "rocprof pmc failed"; tail -20 "$R/gpurun_out/pmc_$tag.log"; exit 1; }
python3 - "$R/gpurun_out/pmc_$tag" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv"); sys.exit(1)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    if "fra" not in k: continue
    k = k.split("(")[0][-70:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
disp = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    if "fra" in k: disp[(k.split("(")[0][-70:], r["Dispatch_Id"])] += 1
nd = collections.Counter(k for k, _ in disp)
for k, v in agg.items():
    print(k, "dispatches", nd[k])
    for c, x in sorted(v.items()): print("    %-28s %16.0f per-dispatch %14.0f" % (c, x, x / max(nd[k], 1)))
PY
