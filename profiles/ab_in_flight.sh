#!/bin/bash
# round 5: what the ED pruner's launcher does with the number of queries in flight (kvm_run_jobs tells it; KVM_ED_IN_FLIGHT
# overrides for this experiment) -- the bench's streams, alternating on one box.  "1" = every launch planned as if alone
# (the behaviour before), "" = the library's own count (8 here)
cd "$(dirname "$0")/.."
one() {
  env ${1:+KVM_ED_IN_FLIGHT=$1} ${2:+KVM_ED_B16=$2} python3 bench.py --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rsm_ed : value %.4g  kernel_ms %.4f frac %.3f' % (d['value'], r['kernel_ms'], r['frac']))"
  env ${1:+KVM_ED_IN_FLIGHT=$1} ${2:+KVM_ED_B16=$2} python3 bench.py --workload cnsm_ed --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cnsm_ed: value %.4g  kernel_ms %.4f' % (d['value'], r['kernel_ms']))"
}
for rep in 1 2; do
  echo "== rep $rep default (in flight as the library counts it)"; one "" ""
  echo "== rep $rep planned as if alone (KVM_ED_IN_FLIGHT=1)"; one 1 ""
  echo "== rep $rep planned as if alone, class B on four-lane rows (KVM_ED_IN_FLIGHT=1 KVM_ED_B16=1)"; one 1 1
  echo "== rep $rep KVM_ED_IN_FLIGHT=2"; one 2 ""
  echo "== rep $rep KVM_ED_IN_FLIGHT=4"; one 4 ""
  echo "== rep $rep default, class B 4 per lane (KVM_ED_B16=0)"; one "" 0
done
