#!/bin/bash
# Round 4, last build: the robustness scripts of rounds 2-3 that stress_r04.sh does not run (ED engines, malformed / hostile /
# messy interval lists, exotic parameters, index build, full-scan executors, context lifecycles, spikes, knife-edge thresholds)
# -- the C ABI was split over six translation units and got its exception barrier this round.  Output goes straight to the
# log (a command that is silent for seven minutes is taken to be hung on the GPU box).
cd "$(dirname "$0")/.."
LOG=gpurun_out/r04_stress_rest.log; mkdir -p gpurun_out; : > $LOG
run() { echo "== $*" | tee -a $LOG; timeout -k 10 400 "$@" 2>&1 | grep --line-buffered -v amdgpu.ids | tee -a $LOG | tail -3; echo "rc=${PIPESTATUS[0]}" | tee -a $LOG; }
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
