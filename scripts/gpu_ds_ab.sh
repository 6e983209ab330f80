#!/bin/bash
# pair launches of small batches with and without dual slices (build_ab/rb_ds0.so), conv_4 and the whole batch-1 forward
for n in tree rb_ds0 tree rb_ds0; do
  if [ $n = tree ]; then unset FRA_LIB_PATH; else export FRA_LIB_PATH=$PWD/build_ab/$n.so; fi
  echo "== $n"
  for b in 1 8; do timeout -k 10 60 python scripts/diag/rb_time.py $b 2 3000 || exit 1; done
  timeout -k 10 120 python scripts/diag/batch_sweep.py 1 2 8 32 2>&1 | grep -v amdgpu || exit 1
done
