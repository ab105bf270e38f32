#!/bin/bash
# round 5: the nine window-split pairs on the bench list, class B on the four-lane rows (working tree at the time) against the
# build before it (profiles/variants/libkvmatch_hip_b4.so): stream value and the single-launch kernel time
cd "$(dirname "$0")/.."
for lib in "" profiles/variants/libkvmatch_hip_b4.so; do
 echo "== lib ${lib:-working tree}"
 for p in 1,1 1,2 1,4 2,1 2,2 2,4 4,1 4,2 4,4; do
  KVM_ED_SPLIT=$p KVM_LIB=${lib:+$PWD/$lib} python3 bench.py --steps 3 --warmup 1 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$p value %.4g  kernel_ms %.4f' % (d['value'], r['kernel_ms']))"
 done
done
