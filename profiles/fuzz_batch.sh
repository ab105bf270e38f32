#!/bin/bash
# A batch of fresh seeds through every fuzzer (seconds each); SEED0 in the environment shifts the seeds.
cd "$(dirname "$0")/.."
S=${SEED0:-40}
run() { echo "== $*"; timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300; }
for i in 0 1 2; do
  run python -u profiles/fuzz_lists.py 400 $((S+i))
  run python -u profiles/fuzz_lists.py 300 $((S+10+i)) 30 hostile
  run python -u profiles/fuzz_lists.py 100 $((S+20+i)) 1500
  run python -u profiles/fuzz_lists.py 40 $((S+30+i)) 8 wide
  run python -u profiles/fuzz_spikes.py 200 $((S+40+i))
  run python -u profiles/fuzz_index.py 40 $((S+50+i))
  run python -u profiles/fuzz_ucr.py 30 $((S+60+i))
  run python -u profiles/fuzz_lifecycle.py 300 $((S+70+i))
done
