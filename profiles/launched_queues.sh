#!/bin/bash
# round 5: the launched path (one RCCL rank) -- exchange stream at normal / high priority, 4 (default) / 8 / 16 hardware queues,
# 8 / 16 slots; the host path beside it
cd "$(dirname "$0")/.."
p=29560
one() {  # prio hwq slots
  p=$((p+1))
  env KVM_COMM_STREAM_PRIORITY=$1 ${2:+GPU_MAX_HW_QUEUES=$2} python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 1 --in-flight $3 --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 --sub-workloads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('launched: exchange stream $1, hw queues ${2:-default}, slots $3: value %.4g early abandon %.4g' % (d['value'], d.get('value_early_abandon') or 0))"
}
for slots in 8 16; do
  one normal "" $slots; one high "" $slots; one normal 8 $slots; one high 8 $slots; one high 16 $slots
done
for q in "" 8 16; do
env ${q:+GPU_MAX_HW_QUEUES=$q} python3 bench.py --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 --sub-workloads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('host path, hw queues ${q:-default}, in flight 8: value %.4g early abandon %.4g' % (d['value'], d.get('value_early_abandon') or 0))"
done
