#!/bin/bash
# HBM traffic of the search kernel: PMC passes (FETCH_SIZE; L2 hits / misses) over a short search-only bench run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
rows=${1:-1000000}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf "$R/gpurun_out/pmc_search_$tag"
  timeout -s KILL 180 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_search_$tag" -- python3 "$R/bench.py" --search-only --gallery-rows $rows --search-steps 2 > "$R/gpurun_out/pmc_search_$tag.log" 2>&1 || { echo "pass $tag failed"; tail -5 "$R/gpurun_out/pmc_search_$tag.log"; exit 1; }
done
python3 - "$R" <<'PY'
import csv, glob, sys, collections
R = sys.argv[1]
for d in glob.glob(f"{R}/gpurun_out/pmc_search_*/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        tot = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "search_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
        for c, dd in tot.items():
            vals = sorted(sum(v) for v in dd.values())
            print(c, "dispatches", len(vals), "max per dispatch %.4g" % vals[-1], "all", ["%.3g" % v for v in vals[-6:]])
PY
