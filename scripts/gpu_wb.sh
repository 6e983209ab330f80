#!/bin/bash
# First checks of the wide-wave chain kernel: the chain tests, then its time beside the eight-wave kernel's.
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -k "rblock_chain" -x -q --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/wb_tests.log 2>&1
rc=$?; tail -n 15 gpurun_out/wb_tests.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
for b in 1024 512 256 128 64; do
  FRA_WBLOCK=1 timeout -k 10 120 python scripts/diag/rb_time.py $b 6 200 || exit 1
  FRA_WBLOCK=0 timeout -k 10 120 python scripts/diag/rb_time.py $b 6 200 || exit 1
done
