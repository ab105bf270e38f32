#!/bin/bash
# same-box A/B of the lane-granular ED pruner (kvm_ed_g.hip): class A/B/C tiling of round 2 (KVM_ED_GEN=0) against the new kernel
# with and without the half-chunk stagger, one and four queries in flight
cd "$(dirname "$0")/.."
line() { python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 in-flight', d['config']['queries_in_flight'], 'value %.4e ms/query %.4f kernel_ms %.4f frac %.3f' % (d['value'], d['ms_per_query'], d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for F in 1 4; do
  KVM_ED_GEN=0 python3 bench.py --steps 10 --warmup 3 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | line "A/B/C tiling (round 2)   "
  KVM_EDG_STAGGER=0 python3 bench.py --steps 10 --warmup 3 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | line "lane-granular, in step   "
  python3 bench.py --steps 10 --warmup 3 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | line "lane-granular, staggered "
done
