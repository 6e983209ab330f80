#!/bin/bash
# usage: gpu_stamps.sh "<lib1> <lib2>" n_ops...
set -e
libs=$1; shift
for l in $libs; do
cp $l face_recognition_amd/libface_recognition_amd.so
for n in "$@"; do echo "=== $l n_ops $n"; timeout -k 10 120 python3 scripts/diag/run_block_stamps.py $n; done
done
