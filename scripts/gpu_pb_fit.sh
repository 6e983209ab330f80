#!/bin/bash
# conv_3 as the pipeline (FRA_PBLOCK_MIN=1) and as bands (FRA_PBLOCK_MIN=0) over batch sizes: gpu_pb_fit.sh <batch sizes...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
for m in 1 0; do
  echo "== FRA_PBLOCK_MIN=$m"
  FRA_PBLOCK_MIN=$m SWEEP_DETAIL=1 timeout -k 10 300 python -u $R/scripts/diag/batch_sweep.py "$@" | sed "s/faces.s  |.*conv_23 [0-9]*//; s/conv_34.*//" || exit 1
done
