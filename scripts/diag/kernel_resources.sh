#!/bin/bash
# Registers, spills and LDS per kernel of one source: kernel_resources.sh <csrc file stem> [extra hipcc flags]
here=$(cd "$(dirname "$0")/../.." && pwd)
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -fno-slp-vectorize "$@" \
    -Rpass-analysis=kernel-resource-usage -c "$here/face_recognition_amd/csrc/$f.hip" -o /dev/null 2>&1 |
  python3 -c '
import re, sys, subprocess
name = None; row = {}
def flush():
    if name: print("%-100s VGPR %3s spill %3s scratch %4s" % (name[:100], row.get("VGPRs"), row.get("VGPRs Spill"), row.get("ScratchSize [bytes/lane]")))
for line in sys.stdin:
    m = re.search(r"remark:\s+(?:Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
    if m:
        flush(); row = {}
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void fra::", "")
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m: row[m.group(1).strip()] = m.group(2)
flush()'
