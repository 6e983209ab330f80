#!/bin/bash
# A variant of the wide-wave chain kernel for A/B runs: wb_build.sh <name> [-DFLAG ...]
# compiles csrc/kernels_wblock.hip with the extra flags and links it with the tree's other objects into
# build_ab/wb_<name>.so (run face_recognition_amd.build first); FRA_LIB_PATH=<that file> makes the package load it.
set -e
name=$1; shift
here=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$here/build_ab"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -Wall -Wno-unused-function -fno-slp-vectorize \
    -mllvm -amdgpu-mfma-vgpr-form "$@" -c "$here/face_recognition_amd/csrc/kernels_wblock.hip" -o "$here/build_ab/wb_$name.o"
objs=$(ls "$here"/face_recognition_amd/build/*.o | grep -v "kernels_wblock.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$here/build_ab/wb_$name.so" $objs "$here/build_ab/wb_$name.o"
echo "$here/build_ab/wb_$name.so"
