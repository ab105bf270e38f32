#!/bin/bash
# A16 with the size rule: ED parity tests, then A/B on one box (KVM_ED_NO_A16=1 = class A on 8 per lane everywhere)
set -o pipefail
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py tests/test_golden.py tests/test_adversarial_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "not dtw" > gpurun_out/r05_a16_tests3.log 2>&1; rc=$?
tail -2 gpurun_out/r05_a16_tests3.log; echo "tests rc $rc"
[ $rc -eq 0 ] || exit $rc
for no in "" 1; do
  echo "== KVM_ED_NO_A16=${no:-unset}"
  for wl in rsm_ed cnsm_ed; do
    KVM_ED_NO_A16=$no python3 bench.py --workload $wl --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$wl: value %.4g  abandon %.4g  kernel_ms %.4f frac %.3f' % (d['value'], d.get('value_early_abandon') or 0, r['kernel_ms'], r['frac']), {k: r.get(k) for k in ('list_bound_ms','full_tiles_kernel_ms')})"
  done
done
