#!/bin/bash
# round 5: the early-abandoning streams are bound by how many call chains overlap (the kernel trace of the host path shows at
# most 4 kernels at a time = the runtime's 4 hardware queues; a chain is ~200 us of which ~140 the pruner whose length is the
# in-order exact sum of the answers): queries in flight x hardware queues, one box
cd "$(dirname "$0")/.."
for wl in rsm_ed cnsm_ed; do for f in 8 16; do for q in "" 8 16; do
env ${q:+GPU_MAX_HW_QUEUES=$q} python3 bench.py --workload $wl --in-flight $f --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 --sub-workloads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl in flight $f hw queues ${q:-4 (default)}: value %.4g early abandon %.4g' % (d['value'], d.get('value_early_abandon') or 0))"
done; done; done
