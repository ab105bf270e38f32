#!/bin/bash
# round 5: do the streams gain from more than 8 queries in flight?  (values per depth, one box: the headline -- every point
# evaluated -- and the same stream with the reference's early abandoning, which is bound by its launch chain)
cd "$(dirname "$0")/.."
for f in ${DEPTHS:-2 4 8 12 16}; do
  python3 bench.py --steps 5 --warmup 2 --in-flight $f --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('in flight $f: value %.4g  ms_per_step %.2f kernel_ms %.4f   early abandon %.4g' % (d['value'], d['ms_per_step'], r['kernel_ms'], d.get('value_early_abandon') or 0))"
done
for f in ${DEPTHS:-2 4 8 12 16}; do
  python3 bench.py --workload cnsm_ed --steps 5 --warmup 2 --in-flight $f --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cnsm_ed in flight $f: value %.4g   early abandon %.4g' % (d['value'], d.get('value_early_abandon') or 0))"
done
