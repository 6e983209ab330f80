#!/bin/bash
# Issue-side counters per kernel of the forward: one PMC pass (8 SQ counters) over `bench.py --steps 2 --warmup 1`.
# Usage: gpu_sq_counters.sh [counters...] (at most 8 SQ_ counters per pass).  Prints, per kernel, each counter's per-dispatch average (summed over the chip's shader engines as rocprofv3 reports it).
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
C=${*:-SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS}
timeout -s KILL 180 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_sq" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-search --no-cpu --no-latency > "$R/gpurun_out/pmc_sq.log" 2>&1 || { echo "rocprof failed"; tail -20 "$R/gpurun_out/pmc_sq.log"; exit 1; }
python3 - "$R" <<'PY'
import csv, glob, sys, collections
R = sys.argv[1]
f = glob.glob(f"{R}/gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "fra" not in k: continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k in tot:
    n = len(disp[k])
    print(k[:70])
    print("   " + "  ".join("%s %.3g" % (c.replace("SQ_", ""), v / n) for c, v in sorted(tot[k].items())))
PY
