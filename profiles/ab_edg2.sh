#!/bin/bash
# what the lane-granular pruner's time is made of (same box): the whole kernel, without the LDS-DMA of all but a pass's first chunk
# (KVM_EDG_DBG=1: wrong answers, timing only), with more / fewer wavefronts than wave slots
cd "$(dirname "$0")/.."
line() { python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 kernel_ms %.4f value %.4e' % (d['roofline']['kernel_ms'], d['value']))"; }
KVM_TRACE=1 python3 bench.py --steps 3 --warmup 1 --extras 0 --cpu-baseline 0 --in-flight 1 2>&1 | grep 'ed_g' | sort | uniq -c | head -5
python3 bench.py --steps 6 --warmup 2 --extras 0 --cpu-baseline 0 --in-flight 1 2>/dev/null | line "full          "
KVM_EDG_DBG=1 python3 bench.py --steps 6 --warmup 2 --extras 0 --cpu-baseline 0 --in-flight 1 2>/dev/null | line "no DMA        "
KVM_EDG_WAVES=4096 python3 bench.py --steps 6 --warmup 2 --extras 0 --cpu-baseline 0 --in-flight 1 2>/dev/null | line "4096 waves    "
KVM_EDG_WAVES=1024 python3 bench.py --steps 6 --warmup 2 --extras 0 --cpu-baseline 0 --in-flight 1 2>/dev/null | line "1024 waves    "
