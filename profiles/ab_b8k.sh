#!/bin/bash
# round 5: class B on the four-lane rows (8 per lane, Karatsuba; the working tree) against class B on the old form
# (profiles/variants/libkvmatch_hip_b4.so = the build before that change), alternating on one box: the headline stream,
# the single-launch kernel time on the bench list, and the cNSM-ED stream
cd "$(dirname "$0")/.."
one() {
  KVM_LIB=${1:+$PWD/$1} python3 bench.py --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rsm_ed : value %.4g  kernel_ms %.4f frac %.3f list_bound_ms %.4f' % (d['value'], r['kernel_ms'], r['frac'], r['list_bound_ms']))"
  KVM_LIB=${1:+$PWD/$1} python3 bench.py --workload cnsm_ed --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cnsm_ed: value %.4g  kernel_ms %.4f' % (d['value'], r['kernel_ms']))"
}
for rep in 1 2 3; do
  echo "== rep $rep class B 8 per lane on four-lane rows"; one ""
  echo "== rep $rep class B 4 per lane (before)"; one profiles/variants/libkvmatch_hip_b4.so
done
