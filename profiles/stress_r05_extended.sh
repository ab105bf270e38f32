#!/bin/bash
# Round 5, end: the parity scripts once more with other seeds and larger counts, on the last build -- ED scripts as the launcher
# decides and with the throughput plan on every call (KVM_ED_IN_FLIGHT=8), DTW / adversarial scripts with fresh seeds.
cd "$(dirname "$0")/.."
LOG=gpurun_out/r05_stress_extended.log; mkdir -p gpurun_out; : > $LOG
run() { echo "== $*" | tee -a $LOG; timeout -k 10 500 "$@" 2>&1 | grep --line-buffered -v amdgpu.ids | tee -a $LOG | tail -2; echo "rc=${PIPESTATUS[0]}" | tee -a $LOG; }
for plan in 0 8; do
  if [ $plan = 8 ]; then export KVM_ED_IN_FLIGHT=8; echo "#### throughput plan on every call (KVM_ED_IN_FLIGHT=8)" | tee -a $LOG; fi
  run python -u profiles/stress_ed.py 600
  run python -u profiles/fuzz_lists.py 1200 17
  run python -u profiles/fuzz_lists.py 300 23 1500
  run python -u profiles/fuzz_lists.py 500 41 30 hostile
  run python -u profiles/fuzz_spikes.py 300 11
  run python -u profiles/fuzz_knife.py 100 13 12
done
unset KVM_ED_IN_FLIGHT
DP_FORM=0 SEED=6101 run python -u profiles/stress_adversarial.py 3
DP_FORM=3 SEED=6102 run python -u profiles/stress_adversarial.py 3
DP_FORM=2 SEED=6103 run python -u profiles/stress_adversarial.py 2
SEED=6104 run python -u profiles/stress_dtw.py 150
run python -u profiles/fuzz_ucr.py 90 9
run python -u profiles/fuzz_lifecycle.py 800 11
