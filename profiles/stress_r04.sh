#!/bin/bash
# Round 4 parity campaign on the final build: the adversarial sweep (ties, sigma = 0, NaN / Inf, bursts, integer data) with every
# DTW list through the packed DP and through the default choice, and random DTW verifications through all four DP forms.
cd "$(dirname "$0")/.."
# (no pipe into tail: a command that prints nothing for seven minutes is taken to be hung on the GPU box)
run() { echo "== $*"; timeout -k 10 500 "$@" 2>&1 | grep --line-buffered -v "amdgpu.ids\|^  huge\|^round"; }
PART=${1:-all}
if [ $PART = all ] || [ $PART = a ]; then
DP_FORM=3 SEED=11 run python -u profiles/stress_adversarial.py 3
DP_FORM=3 SEED=2025 run python -u profiles/stress_adversarial.py 2
fi
if [ $PART = all ] || [ $PART = b ]; then
DP_FORM=0 SEED=404 run python -u profiles/stress_adversarial.py 2
run python -u profiles/stress_dtw.py 60
SEED=77 run python -u profiles/stress_dtw.py 40
SEED=1234 run python -u profiles/stress_dtw.py 40
fi
if [ $PART = c ]; then   # (the last build of the round: other seeds)
DP_FORM=0 SEED=9001 run python -u profiles/stress_adversarial.py 3
DP_FORM=3 SEED=9002 run python -u profiles/stress_adversarial.py 2
SEED=9003 run python -u profiles/stress_dtw.py 60
SEED=9004 run python -u profiles/stress_dtw.py 40
fi
if [ $PART = d ]; then   # (more seeds on the last build: the campaign, not the suite, found the row-bound tie)
DP_FORM=0 SEED=31337 run python -u profiles/stress_adversarial.py 2
DP_FORM=3 SEED=424242 run python -u profiles/stress_adversarial.py 2
DP_FORM=2 SEED=777 run python -u profiles/stress_adversarial.py 2
DP_FORM=0 SEED=5150 run python -u profiles/stress_adversarial.py 2
SEED=8086 run python -u profiles/stress_dtw.py 60
SEED=6502 run python -u profiles/stress_dtw.py 60
fi
