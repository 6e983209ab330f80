#!/bin/bash
# build build_ab/lib_stamps<T>.so for the given DBG_T thread ids
set -e
cd "$(dirname "$0")/../.."
python scripts/diag/make_block_stamps.py
B=face_recognition_amd/build
for T in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -fno-slp-vectorize -DDBG_T=$T -Iinclude -Iface_recognition_amd/csrc -c build_ab/kb_stamps.hip -o build_ab/kbs$T.o
  objs=$(ls $B/*.o | grep -v "/kernels_block.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_ab/lib_stamps$T.so $objs build_ab/kbs$T.o
done
ls build_ab/*.so
