#!/bin/bash
# Build an alternative library for gpu_ab.sh: build_alt.sh <variant source> <name> [replaced source, default kernels_block]
# compiles the variant in place of csrc/<replaced>.hip and links it with the tree's other objects
# into build_ab/<name>.so (run face_recognition_amd.build first so those objects are current).
set -e
src=$1; name=$2; which=${3:-kernels_block}
here=$(cd "$(dirname "$0")/.." && pwd)
cp "$src" "$here/face_recognition_amd/csrc/_alt_$name.hip"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -Wall -Wno-unused-function -fno-slp-vectorize \
    -c "$here/face_recognition_amd/csrc/_alt_$name.hip" -o "$here/build_ab/$name.o"
rm "$here/face_recognition_amd/csrc/_alt_$name.hip"
objs=$(ls "$here"/face_recognition_amd/build/*.o | grep -v "$which.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$here/build_ab/$name.so" $objs "$here/build_ab/$name.o"
echo "$here/build_ab/$name.so"
