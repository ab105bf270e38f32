#!/bin/bash
# One-off robustness sweep of round 2 (after the DTW cascade was rebuilt): every stress / edge script against the oracle.
cd "$(dirname "$0")/.."
run() { echo "== $*"; timeout -k 10 500 "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-4}; echo "rc=$?"; }
run python -u profiles/messy_lists.py
run python -u profiles/exotic_params.py
run python -u profiles/fuzz_index.py 64 2
run python -u profiles/fuzz_ucr.py 45 4
run python -u profiles/fuzz_lifecycle.py 400 5
run python -u profiles/fuzz_spikes.py 150 3
run python -u profiles/fuzz_knife.py 60 7 12
