#!/bin/bash
# further adversarial seeds (each ~2-4 min, oracle-bound)
cd "$(dirname "$0")/.."
for S in ${SEEDS:-5 6 7 8}; do
  echo "== seed $S"; SEED=$S timeout -k 10 600 python -u profiles/stress_adversarial.py 2 2>&1 | grep --line-buffered -v amdgpu.ids | grep --line-buffered "^round\|^verifications\|MISMATCH\|so far" ; echo "rc=$?"
done
