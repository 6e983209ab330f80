#!/bin/bash
# A/B of wide-wave kernel variants: gpu_wb_ab.sh "<name> ..." [batch]   (build_ab/wb_<name>.so; "tree" = the in-tree library, "old" = the eight-wave kernel)
B=${2:-1024}
for n in $1; do
  if [ $n = tree ]; then unset FRA_LIB_PATH; export FRA_WBLOCK=1
  elif [ $n = old ]; then unset FRA_LIB_PATH; export FRA_WBLOCK=0
  else export FRA_LIB_PATH=build_ab/wb_$n.so; export FRA_WBLOCK=1; fi
  echo "== $n"
  timeout -k 10 120 python scripts/diag/wb_diff.py 3 2 | cut -c1-60 || exit 1
  timeout -k 10 120 python scripts/diag/rb_time.py $B 6 300 || exit 1
done
