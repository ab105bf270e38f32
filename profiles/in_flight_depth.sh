#!/bin/bash
# round 5: does the headline stream gain from more than 8 queries in flight?  (value per depth, one box)
cd "$(dirname "$0")/.."
for f in 2 4 8 12 16; do
  python3 bench.py --steps 5 --warmup 2 --in-flight $f --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('in flight $f: value %.4g  ms_per_step %.2f kernel_ms %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms']))"
done
