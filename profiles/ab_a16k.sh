#!/bin/bash
# round 5: the 16-per-lane class A in its Karatsuba form (default) against the plain form (profiles/variants/libkvmatch_hip_k0.so:
# kvm_ed_dot.hip built with -DEDD16_KARATSUBA=0) and against 8 per lane (KVM_ED_NO_A16=1), alternating on one box
cd "$(dirname "$0")/.."
one() {
  KVM_ED_NO_A16=$2 KVM_LIB=${1:+$PWD/$1} MS=8192 LS=64,65536 python3 profiles/ed_run_length.py 2>&1 | grep "run length"
  KVM_ED_NO_A16=$2 KVM_LIB=${1:+$PWD/$1} python3 bench.py --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bench list: value %.4g  kernel_ms %.4f frac %.3f list_bound_ms %.4f full_tiles_ms %.4f' % (d['value'], r['kernel_ms'], r['frac'], r['list_bound_ms'], r['full_tiles_kernel_ms']))"
}
for rep in 1 2; do
  echo "== rep $rep Karatsuba (default)"; one "" ""
  echo "== rep $rep plain 16 per lane"; one profiles/variants/libkvmatch_hip_k0.so ""
  echo "== rep $rep 8 per lane"; one "" 1
done
