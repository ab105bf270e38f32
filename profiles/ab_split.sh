#!/bin/bash
# Same box: the pruner's window-split pair (fA,fB) forced through KVM_ED_SPLIT against the launcher's own choice, bench list.
cd "$(dirname "$0")/.."
for S in auto 1,1 1,2 1,4 2,2 2,4 4,4; do
  if [ $S = auto ]; then unset KVM_ED_SPLIT; else export KVM_ED_SPLIT=$S; fi
  for F in 1 4; do
    python bench.py --steps 10 --warmup 3 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('split $S in flight $F value %.4e ms/query %.4f kernel_ms %.4f' % (d['value'], d['ms_per_query'], d['roofline']['kernel_ms']))"
  done
done
