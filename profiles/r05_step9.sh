#!/bin/bash
# packed DP (two candidates per wave) without the full-minimum form in the kernel: 2 and 3 waves per SIMD, against the default
for cnt in 16384 49152; do
for lib in "" profiles/variants/libkvmatch_hip_pf2.so profiles/variants/libkvmatch_hip_pf3.so; do
  echo "== DPs $cnt lib ${lib:-default}"
  DP_COUNT=$cnt KVM_LIB=${lib:+$PWD/$lib} timeout -k 10 200 python3 profiles/dp_bench.py 2 3 2>&1 | grep "^form" | awk 'NR%4==0'
done
done
