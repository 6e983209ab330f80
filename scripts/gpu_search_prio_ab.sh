#!/bin/bash
# 10k queries x $rows rows (default 1M, 8M) with the in-tree library and with the libraries named in $1 (build_ab/<name>.so), one box.
mkdir -p gpurun_out
libs=$1; shift
for rows in ${@:-1000000 8000000}; do
  for n in $libs; do
    echo "== rows $rows lib $n"
    lib=""; [ "$n" != tree ] && lib=$PWD/build_ab/$n.so
    FRA_LIB_PATH=$lib timeout -k 10 200 python bench.py --search-only --gallery-rows $rows --search-steps 5 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_batch'], 'ms', r['value'], 'q/s frac', r['roofline']['frac'], r['result_digest'])" || exit 1
  done
done
