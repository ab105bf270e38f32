#!/bin/bash
# The parity scripts once more on the round's last build, other seeds and larger counts than stress_r06.sh: the ED scripts as
# the launcher decides and with the throughput plan on every call, the hand-over rule's script, DTW, executors, lifecycles, index.
cd "$(dirname "$0")/.."
LOG=gpurun_out/r06_stress_extended.log; mkdir -p gpurun_out; : > $LOG
run() { echo "== $*" | tee -a $LOG; timeout -k 10 300 "$@" 2>&1 | grep --line-buffered -v amdgpu.ids | tee -a $LOG | tail -1; echo "rc=${PIPESTATUS[0]}" | tee -a $LOG; }
for pass in 0 1; do
  unset KVM_PY_IN_FLIGHT
  if [ $pass = 1 ]; then export KVM_PY_IN_FLIGHT=8; echo "#### planned for 8 in flight" | tee -a $LOG; fi
  run python -u profiles/stress_ed.py 600
  run python -u profiles/fuzz_lists.py 1200 $((163 + pass))
  run python -u profiles/fuzz_lists.py 300 $((169 + pass)) 1500
  run python -u profiles/fuzz_lists.py 400 $((161 + pass)) 30 hostile
  run python -u profiles/fuzz_spikes.py 300 $((263 + pass))
  run python -u profiles/fuzz_knife.py 100 $((167 + pass)) 12
  run python -u profiles/fuzz_head.py 100 $((170 + pass))
done
unset KVM_PY_IN_FLIGHT
run python -u profiles/fuzz_index.py 160 162
run python -u profiles/fuzz_ucr.py 90 164
run python -u profiles/fuzz_lifecycle.py 800 165
SEED=7604 run python -u profiles/stress_dtw.py 120
grep -c "^rc=0" $LOG; grep -v "^rc=0" $LOG | grep "^rc=" | sort | uniq -c
