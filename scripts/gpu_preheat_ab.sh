#!/bin/bash
# The driver's short command (--steps 20 --warmup 5) with and without the pre-heat, and the long run, one box, fresh process each.
for a in "--preheat-ms 0 --steps 20 --warmup 5" "--preheat-ms 400 --steps 20 --warmup 5" "--preheat-ms 0 --steps 20 --warmup 5" "--preheat-ms 400 --steps 20 --warmup 5" "--preheat-ms 0 --steps 100 --warmup 20" "--preheat-ms 1500 --steps 20 --warmup 5"; do
  echo "== $a"
  timeout -k 10 200 python bench.py $a --no-search --no-cpu --no-latency | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], 'faces/s', r['ms_per_step'], 'ms/step', r['config']['preheat'])" || exit 1
done
