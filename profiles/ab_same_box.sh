#!/bin/bash
# same-box A/B of two builds of the library: bench.py with KVM_LIB pointing at the old build, then the current one, twice
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 in-flight', d['config']['queries_in_flight'], 'value %.4e ms/query %.4f abandon %.4e kernel_ms %.4f frac %.3f' % (d['value'], d['ms_per_query'], d.get('value_early_abandon', 0), d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
  for F in 1 3; do
    KVM_LIB=$PWD/profiles/libkvmatch_hip_old.so python bench.py --steps 20 --warmup 5 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | line old
    python bench.py --steps 20 --warmup 5 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | line new
  done
done
