#!/bin/bash
# HBM traffic per launch of the forward's kernels: two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over
# `bench.py --steps 2 --warmup 1`, summarised into gpurun_out/pmc_traffic.json.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 180 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_$c" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-search --no-cpu --no-latency > "$R/gpurun_out/pmc_$c.log" 2>&1 || { echo "rocprof $c failed"; tail -20 "$R/gpurun_out/pmc_$c.log"; exit 1; }
done
python3 - "$R" <<'PY'
import csv, glob, sys, json, collections, re
R = sys.argv[1]
sys.path.insert(0, R)
def collect(c):
    f = glob.glob(f"{R}/gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"]); disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: tot[k] / len(disp[k]) for k in tot}
fe, wr = collect("FETCH_SIZE"), collect("WRITE_SIZE")
def label(k):
    # demangled or mangled kernel name -> the label bench.py prints
    m = re.search(r"rblock_kernelIDF16_Li(\d+)ELi(\d+)ELi(\d+)E", k)
    if m: return "rblock_kernel<_Float16, %s, %s, %s>" % m.groups()
    m = re.search(r"pblock_kernelIDF16_Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", k)
    if m: return "pblock_kernel<_Float16, %s, %s, %s, %s>" % m.groups()
    m = re.search(r"[^rp]block_kernelIDF16_Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", k)
    if m: return "block_kernel<_Float16, %s, %s, %s, %s, %s>" % m.groups()
    for key, lab in (("stemdw_kernelIDF16_", "stemdw_kernel<_Float16>"), ("pw_gemm_kernelIDF16_Li128ELi128ELi2E", "pw_gemm_kernel<_Float16, 128, 128, 2>"),
                     ("gdconv7_kernelIDF16_", "gdconv7_kernel<_Float16>"), ("tail_kernelIDF16_", "tail_kernel<_Float16>"), ("linear_l2norm_kernelIDF16_Li16ELi1E", "linear_l2norm_kernel<_Float16, 16, 1>")):
        if key in k: return lab
    return None
out = {}
for k in fe:
    lab = label(k)
    if not lab or k not in wr: continue
    out[lab] = {"FETCH_SIZE_KB": round(fe[k]), "WRITE_SIZE_KB": round(wr[k]), "hbm_bytes_per_launch": int((2 * fe[k] + wr[k]) * 1024)}
doc = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of `bench.py --steps 2 --warmup 1` (B=1024, f16), per-dispatch averages. Counters are in KB; per the MI355X guide FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950, so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. The correction is calibrated for 16-byte-per-lane streams; these kernels also issue 8-byte accesses and fragment-shaped loads, so treat it as +-10%.",
       "kernels": out}
json.dump(doc, open(f"{R}/gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in out.items(): print("%-50s %12d bytes/launch" % (k, v["hbm_bytes_per_launch"]))
PY
