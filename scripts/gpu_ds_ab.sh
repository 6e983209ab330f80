#!/bin/bash
# small-batch band launches: the in-tree library against variants (build_ab/<name>.so): gpu_ds_ab.sh "<name> ..."
for n in ${1:-tree rb_ds0 tree rb_ds0}; do
  if [ $n = tree ]; then unset FRA_LIB_PATH; else export FRA_LIB_PATH=$PWD/build_ab/$n.so; fi
  echo "== $n"
  timeout -k 10 200 python -u -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "rblock_chain" --timeout 120 --timeout-method thread -p no:cacheprovider 2>&1 | tail -1 || exit 1
  for b in 1 8; do timeout -k 10 60 python scripts/diag/rb_time.py $b 2 3000 || exit 1; done
  timeout -k 10 120 python scripts/diag/batch_sweep.py 1 2 8 32 2>&1 | grep -v amdgpu || exit 1
done
