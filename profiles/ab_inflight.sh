#!/bin/bash
# same-box A/B: queries in flight, host API and sharded (launched) path
cd "$(dirname "$0")/.."
python -u profiles/two_streams.py 2>&1 | grep threads
for F in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{\"metric')][-1]); print('host api   in-flight', d['config']['queries_in_flight'], 'value %.4e' % d['value'], 'ms/query %.4f' % d['ms_per_query'], 'abandon %.4e' % d['value_early_abandon'], 'kernel_ms %.4f frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
for F in 1 2 3; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2965$F bench.py --gpus 1 --steps 10 --warmup 3 --extras 0 --cpu-baseline 0 --in-flight $F 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{\"metric')][-1]); print('sharded    in-flight', d['config']['queries_in_flight'], 'value %.4e' % d['value'], 'ms/query %.4f' % d['ms_per_query'], 'abandon %.4e' % d['value_early_abandon'], 'kernel_ms %.4f frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
