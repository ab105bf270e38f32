#!/bin/bash
# round 5: with >= 4 queries in flight, lists of two rounds of the launch's share of the wave slots and more are not split at all
# (default) against the model's pair on the share (KVM_ED_NO_SHARE_RULE=1): the bench's streams, alternating on one box
cd "$(dirname "$0")/.."
one() {
  for wl in rsm_ed; do
  env ${1:+KVM_ED_NO_SHARE_RULE=$1} python3 bench.py --workload $wl --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$wl no_share_rule=${1:-0}: value %.4g  kernel_ms %.4f' % (d['value'], r['kernel_ms']))"
  done
}
for rep in 1 2 3 4; do one ""; one 1; done
