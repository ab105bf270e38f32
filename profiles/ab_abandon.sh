#!/bin/bash
# same-box A/B of build variants (profiles/variants/*.so) on the bench list: the stream's rate with every point evaluated and
# with the reference's early abandoning, one and five queries in flight
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-20s in-flight %d  every point %.4e /s  early abandon %.4e /s' % ('$1', d['config']['queries_in_flight'], d['value'], d['value_early_abandon']))"; }
for rep in 1 2; do
  for F in 1 5; do
    python bench.py --steps 24 --warmup 2 --queries-per-step 100 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | line current
    for v in "$@"; do KVM_LIB=$PWD/profiles/variants/$v python bench.py --steps 24 --warmup 2 --queries-per-step 100 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | line $v; done
  done
done
