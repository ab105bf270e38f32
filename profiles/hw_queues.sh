#!/bin/bash
# round 5: the runtime's number of hardware queues per process (GPU_MAX_HW_QUEUES, default 4) under the host-API streams of
# 8 queries in flight: alternating on one box
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for q in "" 8; do
for wl in rsm_ed cnsm_ed; do
env ${q:+GPU_MAX_HW_QUEUES=$q} python3 bench.py --workload $wl --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 --sub-workloads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep $wl hw queues ${q:-default}: value %.4g early abandon %.4g' % (d['value'], d.get('value_early_abandon') or 0))"
done; done; done
env GPU_MAX_HW_QUEUES=8 python3 bench.py --workload cnsm_dtw --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 --sub-workloads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cnsm_dtw hw queues 8: value %.4g early abandon %.4g' % (d['value'], d.get('value_early_abandon') or 0))"
python3 bench.py --workload cnsm_dtw --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 --sub-workloads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cnsm_dtw hw queues default: value %.4g early abandon %.4g' % (d['value'], d.get('value_early_abandon') or 0))"
