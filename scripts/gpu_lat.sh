#!/bin/bash
# Batch-1 (or $1) forward: per-kernel event times, then a kernel trace of the graph replay (start-to-start gaps).
B=${1:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
timeout -k 10 120 python scripts/diag/lat_profile.py $B || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/lat_trace"
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/lat_trace" -- python3 "$R/scripts/diag/lat_trace.py" $B > "$R/gpurun_out/lat_trace.log" 2>&1 || { tail -5 "$R/gpurun_out/lat_trace.log"; exit 1; }
python3 - "$R" <<'PY'
import csv, glob, sys
R = sys.argv[1]
f = glob.glob(f"{R}/gpurun_out/lat_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last eager forward: from the last stemdw kernel on
starts = [i for i, r in enumerate(rows) if "stemdw" in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-60s start %7.1f us  dur %6.1f us  gap %5.1f us  grid %s wg %s" % (r["Kernel_Name"][:60].replace("void fra::", ""), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?"), r.get("Workgroup_Size_X", "?")))
    prev_end = e
print("total %.1f us" % ((prev_end - t0) / 1e3))
PY
