#!/bin/bash
# builds scripts/diag/libpeaks.so (gfx950) for scripts/diag/peaks.py
set -e
here=$(cd "$(dirname "$0")" && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -o "$here/libpeaks.so" "$here/peaks.hip"
echo "$here/libpeaks.so"
