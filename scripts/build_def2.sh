#!/bin/bash
# Like build_def.sh, but the definitions apply to BOTH fused block sources (kernels_block, kernels_rblock):
#   build_def2.sh <name> -DFOO ...   ->  build_ab/<name>.so
set -e
name=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$here/build_ab"
for which in kernels_block kernels_rblock; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -Wall -Wno-unused-function -fno-slp-vectorize "$@" \
      -c "$here/face_recognition_amd/csrc/$which.hip" -o "$here/build_ab/${name}_$which.o" &
done
wait
objs=$(ls "$here"/face_recognition_amd/build/*.o | grep -v "/kernels_block.o" | grep -v "/kernels_rblock.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$here/build_ab/$name.so" $objs "$here/build_ab/${name}_kernels_block.o" "$here/build_ab/${name}_kernels_rblock.o"
echo "$here/build_ab/$name.so"
