#!/bin/bash
# The search kernel as eight waves (FRA_SEARCH_WIDE=0) and as four wide waves (1): facebank tests with the wide kernel, then
# 10k queries x $1 rows (default 1M and 8M) on both.
mkdir -p gpurun_out
FRA_SEARCH_WIDE=1 timeout -k 10 300 python -u -m pytest tests/test_gpu_facebank.py tests/test_zz_no_alloc.py -x -q -m gpu --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/search_wide_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/search_wide_tests.log
[ $rc -ne 0 ] && exit $rc
for rows in ${@:-1000000 8000000}; do
  for w in 0 1 0 1; do
    echo "== rows $rows wide $w"
    FRA_SEARCH_WIDE=$w timeout -k 10 200 python bench.py --search-only --gallery-rows $rows --search-steps 5 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_batch'], 'ms', r['value'], 'q/s frac', r['roofline']['frac'], r['bruteforce_check'], r['result_digest'])" || exit 1
  done
done
