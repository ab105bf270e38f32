#!/bin/bash
# One-off robustness sweep of round 2 (after the DTW cascade was rebuilt): every stress / edge script against the oracle.
cd "$(dirname "$0")/.."
run() { echo "== $*"; timeout -k 10 500 "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-4}; echo "rc=$?"; }
SEED=11 run python -u profiles/stress_adversarial.py 3
SEED=2024 run python -u profiles/stress_adversarial.py 2
run python -u profiles/stress_dtw.py 60
SEED=77 run python -u profiles/stress_dtw.py 40
