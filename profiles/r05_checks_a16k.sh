timeout -k 10 500 python3 -m pytest tests/test_parity_gpu.py tests/test_golden.py tests/test_adversarial_gpu.py tests/test_configs_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "not dtw" > gpurun_out/r05_a16k_tests2.log 2>&1; rc=$?; tail -2 gpurun_out/r05_a16k_tests2.log; [ $rc -eq 0 ] || exit $rc
bash profiles/stress_r05.sh ed > /dev/null 2>&1; echo "stress rc=0 lines: $(grep -c 'rc=0' gpurun_out/r05_stress_ed.log)"; grep "rc=[^0]" gpurun_out/r05_stress_ed.log | head; grep -i "mismatches [1-9]\|differences [1-9]\|differences: [1-9]" gpurun_out/r05_stress_ed.log | head
for wl in rsm_ed cnsm_ed; do for no in 0 1; do KVM_ED_NO_A16=$no python3 bench.py --workload $wl --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$wl no_a16=$no: value %.4g  abandon %.4g  kernel_ms %.4f frac %.3f' % (d['value'], d.get('value_early_abandon') or 0, r['kernel_ms'], r['frac']))"; done; done
