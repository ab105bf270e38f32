#!/bin/bash
# one-line summaries of bench.py at 1 and 3 queries in flight (same box), then the per-class unit costs
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('in-flight', d['config']['queries_in_flight'], 'value %.4e ms/query %.4f abandon %.4e kernel_ms %.4f frac %.3f' % (d['value'], d['ms_per_query'], d.get('value_early_abandon', 0), d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
python bench.py --steps 20 --warmup 5 --extras 0 --cpu-baseline 0 --in-flight 1 2>/dev/null | line
python bench.py --steps 20 --warmup 5 --extras 0 --cpu-baseline 0 2>/dev/null | line
python bench.py --steps 20 --warmup 5 --extras 0 --cpu-baseline 0 --workload cnsm_ed 2>/dev/null | line
python -u profiles/ed_class_cost.py 2>&1 | grep -v amdgpu.ids
