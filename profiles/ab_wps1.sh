#!/bin/bash
# round 5 (experiment, negative; the two macros are not in the source any more: `__launch_bounds__(256, EDD_WPS)` on
# kvm_ed_dot_kernel and EDD16_AHEAD = 8 allowed by its static_assert): the 16-per-lane class A with window points fetched 8 blocks ahead in up to 512 registers = ONE wave per
# SIMD (profiles/variants/libkvmatch_hip_a8_w1.so: -DEDD16_AHEAD=8 -DEDD_WPS=1: 256 VGPRs + 20 AGPRs, no scratch), and 4 blocks
# ahead without the two-waves register cap (a4_w1: 238 VGPRs, still two waves), against the default build (2 blocks ahead, two
# waves per SIMD): full isolated tiles, long runs, and the bench list under forced window splits (the launcher's model
# assumes 8 wave slots per CU)
cd "$(dirname "$0")/.."
one() {
  lib=$1
  KVM_LIB=${lib:+$PWD/$lib} MS=8192 LS=64,65536 python3 profiles/ed_run_length.py 2>&1 | grep "run length"
  for sp in "" 1,1 1,2 1,4 2,4; do
  env ${sp:+KVM_ED_SPLIT=$sp} KVM_LIB=${lib:+$PWD/$lib} python3 bench.py --steps 3 --warmup 1 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bench list split ${sp:-auto}: value %.4g  kernel_ms %.4f frac %.3f full_tiles_ms %.4f' % (d['value'], r['kernel_ms'], r['frac'], r['full_tiles_kernel_ms']))"
  done
}
echo "== default (2 ahead, 2 waves/SIMD)"; one ""
echo "== 8 ahead, 1 wave/SIMD"; one profiles/variants/libkvmatch_hip_a8_w1.so
echo "== 4 ahead, register cap lifted"; one profiles/variants/libkvmatch_hip_a4_w1.so
echo "== default again"; one ""
