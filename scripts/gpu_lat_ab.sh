#!/bin/bash
# block tests + per-kernel times at batch 1 and 8 + the full-batch forward, with whatever library is in the tree (for gpu_ab.sh)
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q --timeout 120 --timeout-method thread -k "fused_block or golden or large_batch or batch" 2>&1 | tail -1
timeout -k 10 120 python scripts/diag/lat_profile.py 1 && timeout -k 10 120 python scripts/diag/lat_profile.py 8 | tail -1
bash scripts/gpu_fwd.sh | grep -E "faces"
