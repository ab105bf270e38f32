#!/bin/bash
cd "$(dirname "$0")/.."
line() { python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 kernel_ms %.4f value %.4e' % (d['roofline']['kernel_ms'], d['value']))"; }
for S in 0 1 2; do
KVM_EDG_STAGGER=$S python3 bench.py --steps 6 --warmup 2 --extras 0 --cpu-baseline 0 --in-flight 1 2>/dev/null | line "stagger $S  in flight 1"
done
KVM_EDG_STAGGER=2 python3 bench.py --steps 6 --warmup 2 --extras 0 --cpu-baseline 0 --in-flight 4 2>/dev/null | line "stagger 2  in flight 4"
KVM_EDG_STAGGER=2 KVM_EDG_DBG=1 python3 bench.py --steps 6 --warmup 2 --extras 0 --cpu-baseline 0 --in-flight 1 2>/dev/null | line "stagger 2 no DMA      "
