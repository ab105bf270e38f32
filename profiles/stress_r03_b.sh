#!/bin/bash
# One-off robustness sweep of round 2 (after the DTW cascade was rebuilt): every stress / edge script against the oracle.
cd "$(dirname "$0")/.."
run() { echo "== $*"; timeout -k 10 500 "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-4}; echo "rc=$?"; }
run python -u profiles/edge_small.py
run python -u profiles/edge_sizes.py
run python -u profiles/edge_clip.py
run python -u profiles/stress_ed.py 200
run python -u profiles/fuzz_lists.py 600 3
run python -u profiles/fuzz_lists.py 150 9 1500
run python -u profiles/fuzz_lists.py 60 21 8 wide
run python -u profiles/fuzz_lists.py 300 31 30 hostile
