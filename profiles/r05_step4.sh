#!/bin/bash
# A16 variants on one box: prefetch distance 4 (default) / 2 / 1 blocks, and the 8-per-lane build
for lib in "" profiles/variants/libkvmatch_hip_ah2.so profiles/variants/libkvmatch_hip_ah1.so profiles/variants/libkvmatch_hip_a8.so; do
  echo "== lib ${lib:-default (A16, 4 blocks ahead)}"
  KVM_LIB=${lib:+$PWD/$lib} MS=8192 LS=64,65536 python3 profiles/ed_run_length.py 2>&1 | grep "run length"
  KVM_LIB=${lib:+$PWD/$lib} python3 bench.py --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bench list: value %.4g  kernel_ms %.4f frac %.3f list_bound_ms %.4f full_tiles_ms %.4f' % (d['value'], r['kernel_ms'], r['frac'], r['list_bound_ms'], r['full_tiles_kernel_ms']))"
done
