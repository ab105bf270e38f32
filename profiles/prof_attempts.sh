#!/bin/bash
# round 5: `rocprofv3 --kernel-trace --stats -- python3 bench.py` N times in a row, native host threads: how many complete?
# (with the enqueue lock the library takes under a profiler since the end of round 5; KVM_SERIALIZE_ENQUEUE=0 = without)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_attempts; rm -rf $OUT; mkdir -p $OUT
ok=0
for i in $(seq 1 ${N:-6}); do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$i -- python3 bench.py --cpu-baseline 0 > $OUT/bench$i.json 2> $OUT/bench$i.err
  rc=$?; echo "attempt $i rc $rc $(python3 -c "
import json,sys
try:
    d=json.loads(open('$OUT/bench$i.json').read().strip().splitlines()[-1]); print('value %.4g abandon %.4g kernel_ms %.4f' % (d['value'], d['value_early_abandon'], d['roofline']['kernel_ms']))
except Exception as e: print('no line')
")"
  [ $rc = 0 ] && ok=$((ok+1))
  rm -rf $OUT/t$i
done
echo "completed $ok of ${N:-6}"
