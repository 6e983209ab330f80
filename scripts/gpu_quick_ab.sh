#!/bin/bash
# fused-block tests + model golden test + short forward bench with whatever library is in the tree (for gpu_ab.sh)
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q --timeout 120 --timeout-method thread -k "fused_block or golden or large_batch" 2>&1 | tail -1
bash scripts/gpu_fwd.sh "$@" | grep -E "faces|block_kernel"
