#!/bin/bash
# A/B of conv_4 chain variants: gpu_rb_ab.sh "<name> ..." [batch]   (build_ab/<name>.so; "tree" = the in-tree library)
B=${2:-1024}
for n in $1; do
  if [ $n = tree ]; then unset FRA_LIB_PATH; else export FRA_LIB_PATH=build_ab/$n.so; fi
  echo "== $n"
  timeout -k 10 120 python scripts/diag/wb_diff.py 5 3 | cut -c1-50 || exit 1
  timeout -k 10 120 python scripts/diag/rb_time.py $B 6 300 || exit 1
done
