#!/bin/bash
# Forward-only bench (per-kernel table) with the in-tree library and with build_ab/<name>.so for every name in $1, one box:
# gpu_fwd_ab.sh "<name> ..." [grep pattern for the kernel lines] [bench args]
names=$1; pat=${2:-kernel}; shift 2
for n in $names; do
  if [ $n = tree ]; then unset FRA_LIB_PATH; else export FRA_LIB_PATH=$PWD/build_ab/$n.so; fi
  echo "== $n"
  bash scripts/gpu_fwd.sh --steps 50 --warmup 10 "$@" | grep -E "faces|$pat" || exit 1
done
