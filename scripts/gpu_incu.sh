#!/bin/bash
# Executed-instruction counts and pipe-busy cycles per kernel of the forward at batch 1024: two rocprofv3 PMC passes of
# 8 SQ counters each over `bench.py --steps 2 --warmup 1` (counters only; no trace domains beside --kernel-trace).
# Writes gpurun_out/incu_counters.json: per kernel, each counter's per-dispatch average summed over the chip.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
A="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
B="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
rm -rf "$R/gpurun_out/pmc_incu_a" "$R/gpurun_out/pmc_incu_b"
timeout -s KILL 180 rocprofv3 --pmc $A --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_incu_a" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-search --no-cpu --no-latency > "$R/gpurun_out/pmc_incu_a.log" 2>&1 || { echo "pass A failed"; tail -20 "$R/gpurun_out/pmc_incu_a.log"; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc $B --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_incu_b" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-search --no-cpu --no-latency > "$R/gpurun_out/pmc_incu_b.log" 2>&1 || { echo "pass B failed"; tail -20 "$R/gpurun_out/pmc_incu_b.log"; exit 1; }
python3 - "$R" <<'PY'
import collections, csv, glob, json, subprocess, sys
R = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for d in ("pmc_incu_a", "pmc_incu_b"):
    for f in glob.glob(f"{R}/gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "fra" not in k:
                continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {}
for k in tot:
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
    out[name] = {c: v / max(1, len(disp[k][c])) for c, v in tot[k].items()}
    out[name]["dispatches_seen"] = max(len(s) for s in disp[k].values())
json.dump(out, open(f"{R}/gpurun_out/incu_counters.json", "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(k[:90])
    print("   " + "  ".join("%s %.3g" % (c.replace("SQ_", ""), x) for c, x in sorted(v.items())))
PY
