#!/bin/bash
# CPU only (no GPU, no library kernels): the two readings of the Java text -- the C oracle and the Python transliteration --
# and the oracle's executor / engine restatements against each other.  A few minutes on one core.
cd "$(dirname "$0")/.."
run() { echo "== $*"; python -W ignore "$@" 2>&1 | tail -3; echo "rc=$?"; }
run profiles/cpu_fuzz_oracle_vs_transliteration.py 250 ${SEED:-11}
run profiles/cpu_fuzz_adversarial_restatements.py 160 ${SEED:-11}
run profiles/cpu_fuzz_index_restatements.py 120 ${SEED:-11}
run profiles/cpu_fuzz_ucr_restatements.py ${SEED:-11} 600
