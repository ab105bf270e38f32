#!/bin/bash
# round 5: the launched path (ShardedStream over RCCL, kvm_run_device_jobs) with ONE rank, by queries in flight, beside the host
# path (QueryStream, kvm_run_jobs) on the same box
cd "$(dirname "$0")/.."
p=29520
for f in 8 12 16; do
  p=$((p+1))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 1 --in-flight $f --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 --sub-workloads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('launched, 1 RCCL rank, in flight $f: value %.4g early abandon %.4g' % (d['value'], d.get('value_early_abandon') or 0))"
done
python3 bench.py --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 --sub-workloads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('host path, in flight 8: value %.4g early abandon %.4g' % (d['value'], d.get('value_early_abandon') or 0))"
