#!/bin/bash
# band launches (dual slices) for batches of more than one round of band workgroups: build_ab/rb_rounds<N>.so against the tree
for n in ${1:-tree rb_rounds2 rb_rounds3 tree}; do
  if [ $n = tree ]; then unset FRA_LIB_PATH; else export FRA_LIB_PATH=$PWD/build_ab/$n.so; fi
  echo "== $n"
  timeout -k 10 120 python scripts/diag/mode_check.py 36 37 64 73 74 109 2>&1 | grep -v amdgpu || exit 1
  timeout -k 10 200 python scripts/diag/batch_sweep.py 36 37 48 64 73 74 90 109 110 128 2>&1 | grep -v amdgpu || exit 1
done
