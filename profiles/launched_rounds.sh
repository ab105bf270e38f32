#!/bin/bash
# round 5: the launched path (one RCCL rank): one round of slots per context (the slot is handed back by the exchange thread
# before its context starts the next query) against two (a second set of buffers: the next query is enqueued behind the one
# under exchange), 8 and 16 contexts; the host path beside it.  Alternating on one box.
cd "$(dirname "$0")/.."
p=29600
one() {  # rounds contexts
  p=$((p+1))
  KVM_BENCH_SLOT_ROUNDS=$1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 1 --in-flight $2 --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 --sub-workloads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('launched: rounds $1, contexts $2: value %.4g early abandon %.4g' % (d['value'], d.get('value_early_abandon') or 0))"
}
for rep in 1 2; do
  one 1 8; one 2 8; one 1 16; one 2 16; one 3 8
  python3 bench.py --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 --sub-workloads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('host path, in flight 8: value %.4g early abandon %.4g' % (d['value'], d.get('value_early_abandon') or 0))"
done
