#!/bin/bash
# same-box A/B: more queries in flight (host API), 20 steps each
cd "$(dirname "$0")/.."
for F in 3 4 5 6 8 3; do
  python bench.py --steps 20 --warmup 5 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{\"metric')][-1]); print('host api   in-flight', d['config']['queries_in_flight'], 'value %.4e' % d['value'], 'ms/query %.4f' % d['ms_per_query'], 'abandon %.4e' % d['value_early_abandon'], 'kernel_ms %.4f frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
