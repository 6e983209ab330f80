#!/bin/bash
# conv_5 workgroup shapes (one image, two or four stacked images per workgroup) over batch sizes: gpu_rb_stack.sh <batch sizes...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
for g in 1 2 4; do
  echo "== FRA_RB_STACK=$g"
  FRA_RB_STACK=$g SWEEP_DETAIL=1 timeout -k 10 200 python -u $R/scripts/diag/batch_sweep.py "$@" || exit 1
done
