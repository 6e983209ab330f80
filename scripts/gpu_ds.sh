#!/bin/bash
# dual-slice band launches: chain tests, then the pair launch alone (batch 1 and 8) and the batch-1 latency lines
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "rblock_chain" --timeout 120 --timeout-method thread -p no:cacheprovider 2>&1 | tail -5 || exit 1
for b in 1 8; do timeout -k 10 60 python scripts/diag/rb_time.py $b 2 2000 || exit 1; done
