#!/bin/bash
# the bench list under forced window splits (KVM_ED_SPLIT=fA,fB) against the launcher's own choice, one query in flight
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('split %-8s kernel_ms %.4f value %.4e' % ('$1', d['roofline']['kernel_ms'], d['value']))"; }
for rep in 1 2; do
  for sp in model 1,1 1,2 1,4 2,2 2,4 4,4; do
    if [ $sp = model ]; then python bench.py --steps 5 --warmup 2 --extras 0 --cpu-baseline 0 --in-flight 1 2>/dev/null | line model
    else KVM_ED_SPLIT=$sp python bench.py --steps 5 --warmup 2 --extras 0 --cpu-baseline 0 --in-flight 1 2>/dev/null | line $sp; fi
  done
done
