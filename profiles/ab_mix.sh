#!/bin/bash
# round 5 (negative, the code is not kept): class-A workgroups spread evenly over the launch order (KVM_ED_MIX=1 in that build)
# against all in front, per split pair, class B in both forms -- the bench list alone and in the stream
cd "$(dirname "$0")/.."
one() {
  env ${1:+KVM_ED_MIX=$1} ${2:+KVM_ED_SPLIT=$2} ${3:+KVM_ED_B16=$3} python3 bench.py --steps 3 --warmup 1 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('mix=$1 split=$2 b16=$3: value %.4g  kernel_ms %.4f frac %.3f' % (d['value'], r['kernel_ms'], r['frac']))"
}
one "" "" ""
for b in "" 1; do for sp in 2,4 4,4 4,2 2,2 1,1 1,4; do one "" $sp $b; one 1 $sp $b; done; done
one "" "" ""
