#!/bin/bash
# A16 as a per-launch decision (after the stale-SCC workaround): per-tile check, ED parity tests, then the two forms of class A
# under every split pair (fit input)
set -o pipefail
DBG_TILES=200 python3 profiles/r05_dbg2.py 2>&1 | grep "^form" | cut -c1-120
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py tests/test_golden.py tests/test_adversarial_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "not dtw" > gpurun_out/r05_a16_tests2.log 2>&1; rc=$?
tail -3 gpurun_out/r05_a16_tests2.log; echo "tests rc $rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python3 profiles/ed_a16_sweep.py > gpurun_out/r05_a16_sweep.log 2>&1; echo "sweep rc $?"
grep -c "^m " gpurun_out/r05_a16_sweep.log
