#!/bin/bash
# A variant of one source of the library for A/B runs: alt_build.sh <source stem> <name> [-DFLAG ...]
# compiles csrc/<stem>.hip with the extra flags and links it with the tree's other objects into build_ab/<name>.so
# (run face_recognition_amd.build first); FRA_LIB_PATH=<that file> makes the package load it.
set -e
stem=$1; name=$2; shift 2
here=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$here/build_ab"
extra=""
[ "$stem" = kernels_wblock ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize \
    $extra "$@" -c "$here/face_recognition_amd/csrc/$stem.hip" -o "$here/build_ab/$name.o"
objs=$(ls "$here"/face_recognition_amd/build/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$here/build_ab/$name.so" $objs "$here/build_ab/$name.o"
echo "$here/build_ab/$name.so"
