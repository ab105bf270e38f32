#!/bin/bash
# same-box A/B of build variants of the library (profiles/variants/*.so, named on the command line) against the current build:
# the bench line with one query in flight (a launch's own duration) and with the default four
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-24s in-flight %d value %.4e ms/query %.4f kernel_ms %.4f frac %.3f' % ('$1', d['config']['queries_in_flight'], d['value'], d['ms_per_query'], d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
  for F in 1 4; do
    python bench.py --steps 6 --warmup 2 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | line current
    for v in "$@"; do KVM_LIB=$PWD/profiles/variants/$v python bench.py --steps 6 --warmup 2 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | line $v; done
  done
done
