#!/bin/bash
# classes A (8 per lane) and B with global loads 8 blocks ahead of their LDS write instead of 4, against the default, alternating
# on one box; the pure class lists of round 3 (profiles/pure_class_lists.py) and the bench list
for rep in 1 2; do
for lib in "" profiles/variants/libkvmatch_hip_d8.so; do
  echo "== rep $rep lib ${lib:-default}"
  for no in "" 1; do
    KVM_ED_NO_A16=$no KVM_LIB=${lib:+$PWD/$lib} python3 bench.py --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('no_a16=${no:-0}: value %.4g  kernel_ms %.4f frac %.3f list_bound_ms %.4f full_tiles_ms %.4f' % (d['value'], r['kernel_ms'], r['frac'], r['list_bound_ms'], r['full_tiles_kernel_ms']))"
  done
done
done
