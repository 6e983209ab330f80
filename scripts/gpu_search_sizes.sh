#!/bin/bash
# un-profiled search timings at several gallery sizes
set -e
for n in "$@"; do
  timeout -k 10 300 python3 bench.py --search-only --gallery-rows $n --search-steps 5 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['gallery_rows'], 'ms', d['ms_per_batch'], 'q/s', d['value'], 'frac', d['roofline']['frac'], 'hit', d['planted_top1_hit_rate'])"
done
