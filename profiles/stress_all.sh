#!/bin/bash
# One-off robustness sweep of round 2 (after the DTW cascade was rebuilt): every stress / edge script against the oracle.
cd "$(dirname "$0")/.."
run() { echo "== $*"; timeout -k 10 500 "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-4}; echo "rc=$?"; }
SEED=11 run python -u profiles/stress_adversarial.py 3
SEED=2024 run python -u profiles/stress_adversarial.py 2
run python -u profiles/stress_dtw.py 60
SEED=77 run python -u profiles/stress_dtw.py 40
run python -u profiles/edge_small.py
run python -u profiles/edge_sizes.py
run python -u profiles/edge_clip.py
run python -u profiles/stress_ed.py 200
run python -u profiles/fuzz_lists.py 600 3
run python -u profiles/fuzz_lists.py 150 9 1500
run python -u profiles/fuzz_lists.py 60 21 8 wide
run python -u profiles/fuzz_lists.py 300 31 30 hostile
run python -u profiles/messy_lists.py
run python -u profiles/exotic_params.py
run python -u profiles/fuzz_index.py 64 2
run python -u profiles/fuzz_ucr.py 45 4
run python -u profiles/fuzz_lifecycle.py 400 5
run python -u profiles/fuzz_spikes.py 150 3
run python -u profiles/fuzz_knife.py 60 7 12
