#!/bin/bash
# Run the GPU test files one after another; stop at the first crash (exit status > 1), keep going
# after ordinary test failures (status 1) so that one call reports on every file.
mkdir -p gpurun_out
overall=0
for f in "$@"; do
  name=$(basename "$f" .py)
  timeout -k 10 420 python -u -m pytest "$f" -v -m gpu --timeout 120 --timeout-method thread -p no:cacheprovider > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc"; tail -n 3 "gpurun_out/$name.log"
  if [ $rc -gt 1 ]; then echo "crash or timeout in $name: stopping"; exit $rc; fi
  if [ $rc -ne 0 ]; then overall=1; fi
done
exit $overall
