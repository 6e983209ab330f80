#!/bin/bash
# Batch sweep with per-kernel detail for the in-tree library and for alternatives: gpu_sweep_ab.sh "<alt.so> ..." <batch sizes...>
alts=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
lib=$R/face_recognition_amd/libface_recognition_amd.so
cp $lib /tmp/lib_a.so
echo "== in-tree"; SWEEP_DETAIL=1 timeout -k 10 200 python -u $R/scripts/diag/batch_sweep.py "$@" || exit 1
for a in $alts; do cp $a $lib; echo "== $a"; SWEEP_DETAIL=1 timeout -k 10 200 python -u $R/scripts/diag/batch_sweep.py "$@" || exit 1; done
cp /tmp/lib_a.so $lib
echo "== in-tree again"; SWEEP_DETAIL=1 timeout -k 10 200 python -u $R/scripts/diag/batch_sweep.py "$@"
