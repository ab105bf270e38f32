#!/bin/bash
# Round 5 parity campaign: the robustness scripts of rounds 2-4 on the build with the 16-per-lane class A -- once as the launcher
# decides (short lists: 8 per lane) and once with the 16-per-lane form FORCED on every list (KVM_ED_A16=1; device-built lists
# through KVM_ED_A16_MIN_UNITS=0 KVM_ED_SPLIT_UNITS=0), that with class B in its old form (KVM_ED_B16=0) and on the four-lane rows
# (KVM_ED_B16=1), and once as the launcher plans with 8 queries in flight (KVM_ED_IN_FLIGHT=8); the DTW sweeps once (their kernels did not change, the host layer did).
cd "$(dirname "$0")/.."
LOG=gpurun_out/r05_stress_${1:-all}.log; mkdir -p gpurun_out; : > $LOG
run() { echo "== $*" | tee -a $LOG; timeout -k 10 400 "$@" 2>&1 | grep --line-buffered -v amdgpu.ids | tee -a $LOG | tail -2; echo "rc=${PIPESTATUS[0]}" | tee -a $LOG; }
PART=${1:-all}
if [ $PART = all ] || [ $PART = ed ]; then
for forced in 0 1 2 3; do
  if [ $forced = 1 ]; then export KVM_ED_A16=1 KVM_ED_A16_MIN_UNITS=0 KVM_ED_SPLIT_UNITS=0 KVM_ED_B16=0; echo "#### 16 per lane forced, class B 4 per lane" | tee -a $LOG; fi
  if [ $forced = 2 ]; then export KVM_ED_B16=1; echo "#### 16 per lane forced, class B on the four-lane rows (8 per lane)" | tee -a $LOG; fi
  if [ $forced = 3 ]; then unset KVM_ED_A16 KVM_ED_A16_MIN_UNITS KVM_ED_SPLIT_UNITS KVM_ED_B16; export KVM_ED_IN_FLIGHT=8; echo "#### as the launcher decides when told that 8 queries are in flight (the throughput plan on every single call)" | tee -a $LOG; fi
  run python -u profiles/edge_small.py
  run python -u profiles/edge_sizes.py
  run python -u profiles/edge_clip.py
  run python -u profiles/stress_ed.py 200
  run python -u profiles/fuzz_lists.py 400 3
  run python -u profiles/fuzz_lists.py 100 9 1500
  run python -u profiles/fuzz_lists.py 200 31 30 hostile
  run python -u profiles/messy_lists.py
  run python -u profiles/exotic_params.py
  run python -u profiles/fuzz_spikes.py 100 3
  run python -u profiles/fuzz_knife.py 40 7 12
done
unset KVM_ED_A16 KVM_ED_A16_MIN_UNITS KVM_ED_SPLIT_UNITS KVM_ED_B16 KVM_ED_IN_FLIGHT
fi
if [ $PART = all ] || [ $PART = rest ]; then
  run python -u profiles/fuzz_index.py 64 2
  run python -u profiles/fuzz_ucr.py 45 4
  run python -u profiles/fuzz_lifecycle.py 400 5
  DP_FORM=0 SEED=5001 run python -u profiles/stress_adversarial.py 2
  DP_FORM=3 SEED=5002 run python -u profiles/stress_adversarial.py 2
  SEED=5003 run python -u profiles/stress_dtw.py 60
fi
