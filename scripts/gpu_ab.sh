#!/bin/bash
# Compare builds of the library on the same box: gpu_ab.sh "<alt1.so> <alt2.so> ..." <command...>
# runs the command with the in-tree library, then with each alternative copied over it, then the
# in-tree one again.
set -e
alts=$1; shift
lib=face_recognition_amd/libface_recognition_amd.so
cp $lib /tmp/lib_a.so
echo "== in-tree"; "$@"
for a in $alts; do cp $a $lib; echo "== $a"; "$@"; done
cp /tmp/lib_a.so $lib
echo "== in-tree again"; "$@"
