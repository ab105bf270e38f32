#!/bin/bash
# Round 6 parity campaign on the final build: the robustness scripts of rounds 2 - 5 (random / hostile / messy lists, edge sizes,
# spikes, knife-edge thresholds; every script compares the library with the oracle bit for bit and prints its mismatches)
#   - as the launcher decides;
#   - with the 16-per-lane form of class A FORCED on every list (device-built cNSM lists of any size through it, never split),
#     class B in its 4-per-lane form, then on the four-lane rows (8 per lane);
#   - as the launcher plans when told that 8 queries are in flight;
# through the binding's KVM_PY_ED_FORCE / KVM_PY_IN_FLIGHT (KVM_OPT_ED_FORCE / KVM_OPT_IN_FLIGHT on every context the scripts open);
# then index, executors, context lifecycles, the adversarial DTW sweeps through the DP forms.   stress_r06.sh [ed|rest|all]
cd "$(dirname "$0")/.."
LOG=gpurun_out/r06_stress_${1:-all}.log; mkdir -p gpurun_out; : > $LOG
run() { echo "== $*" | tee -a $LOG; timeout -k 10 400 "$@" 2>&1 | grep --line-buffered -v amdgpu.ids | tee -a $LOG | tail -2; echo "rc=${PIPESTATUS[0]}" | tee -a $LOG; }
PART=${1:-all}
# KVM_ED_FORCE(split_a, split_b, a16, b16, classes, dev_a16_any, dev_no_split) = a | b<<4 | a16<<8 | b16<<10 | classes<<12 | any<<14 | nosplit<<15
F16_B4=$(( (2<<8) | (1<<10) | (1<<14) | (1<<15) ))
F16_B8=$(( (2<<8) | (2<<10) | (1<<14) | (1<<15) ))
if [ $PART = all ] || [ $PART = ed ]; then
for forced in 0 1 2 3; do
  unset KVM_PY_ED_FORCE KVM_PY_IN_FLIGHT
  if [ $forced = 0 ]; then echo "#### as the launcher decides" | tee -a $LOG; fi
  if [ $forced = 1 ]; then export KVM_PY_ED_FORCE=$F16_B4; echo "#### 16 per lane forced, class B 4 per lane" | tee -a $LOG; fi
  if [ $forced = 2 ]; then export KVM_PY_ED_FORCE=$F16_B8; echo "#### 16 per lane forced, class B on the four-lane rows (8 per lane)" | tee -a $LOG; fi
  if [ $forced = 3 ]; then export KVM_PY_IN_FLIGHT=8; echo "#### as the launcher decides when told that 8 queries are in flight (the throughput plan on every single call)" | tee -a $LOG; fi
  run python -u profiles/edge_small.py
  run python -u profiles/edge_sizes.py
  run python -u profiles/edge_clip.py
  run python -u profiles/stress_ed.py 200
  run python -u profiles/fuzz_lists.py 400 63
  run python -u profiles/fuzz_lists.py 100 69 1500
  run python -u profiles/fuzz_lists.py 200 61 30 hostile
  run python -u profiles/messy_lists.py
  run python -u profiles/exotic_params.py
  run python -u profiles/fuzz_spikes.py 100 63
  run python -u profiles/fuzz_knife.py 40 67 12
  run python -u profiles/fuzz_head.py 30 $((70 + forced))
done
unset KVM_PY_ED_FORCE KVM_PY_IN_FLIGHT
fi
if [ $PART = all ] || [ $PART = rest ]; then
  run python -u profiles/fuzz_index.py 64 62
  run python -u profiles/fuzz_ucr.py 45 64
  run python -u profiles/fuzz_lifecycle.py 400 65
  DP_FORM=0 SEED=6601 run python -u profiles/stress_adversarial.py 2
  DP_FORM=3 SEED=6602 run python -u profiles/stress_adversarial.py 2
  DP_FORM=2 SEED=6603 run python -u profiles/stress_adversarial.py 1
  SEED=6604 run python -u profiles/stress_dtw.py 60
fi
grep -c "^rc=0" $LOG; grep -v "^rc=0" $LOG | grep "^rc=" | sort | uniq -c
