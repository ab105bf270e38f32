#!/bin/bash
# same-box A/B, end of round 3: the default pruner (class A / B / C tiling, class A stream-staged) against the lane-granular
# LDS-DMA pruner (KVM_ED_GEN=1), one and four queries in flight, twice
cd "$(dirname "$0")/.."
line() { python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 in-flight', d['config']['queries_in_flight'], 'value %.4e ms/query %.4f kernel_ms %.4f frac %.3f' % (d['value'], d['ms_per_query'], d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
for F in 1 4; do
  python3 bench.py --steps 8 --warmup 3 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | line "default (stream-staged A/B/C)"
  KVM_ED_GEN=1 python3 bench.py --steps 8 --warmup 3 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | line "lane-granular (LDS-DMA)      "
done
done
