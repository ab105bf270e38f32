#!/bin/bash
# same-box A/B: 3 vs 4 vs 5 queries in flight for every workload
cd "$(dirname "$0")/.."
for W in rsm_ed cnsm_ed cnsm_dtw mixed; do
 for F in 3 4 5 3 4; do
  python bench.py --steps 10 --warmup 3 --extras 0 --cpu-baseline 0 --workload $W --in-flight $F 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$W in-flight', d['config']['queries_in_flight'], 'value %.4e ms/query %.4f abandon %.4e' % (d['value'], d['ms_per_query'], d.get('value_early_abandon', 0)))"
 done
done
