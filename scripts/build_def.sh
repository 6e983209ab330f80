#!/bin/bash
# Build a variant library for gpu_ab.sh from the tree's own sources plus preprocessor definitions:
#   build_def.sh <name> <source stem, e.g. kernels_rblock> -DFOO -DBAR=2 ...
# compiles csrc/<stem>.hip with the definitions and links it with the tree's other objects into build_ab/<name>.so
# (run face_recognition_amd.build first so those objects are current).
set -e
name=$1; which=$2; shift 2
here=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$here/build_ab"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -Wall -Wno-unused-function -fno-slp-vectorize "$@" \
    -c "$here/face_recognition_amd/csrc/$which.hip" -o "$here/build_ab/$name.o"
objs=$(ls "$here"/face_recognition_amd/build/*.o | grep -v "/$which.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$here/build_ab/$name.so" $objs "$here/build_ab/$name.o"
echo "$here/build_ab/$name.so"
