#!/bin/bash
# same-box A/B of the DP kernel: profiles/libkvmatch_hip_old.so (the build before the change) against the current build --
# throughput over 16 384 full DPs (dp_bench.py) and the latency of one DP (dp_lone.py), twice each
cd "$(dirname "$0")/.."
for rep in 1 2; do
  echo "== old"; KVM_LIB=$PWD/profiles/libkvmatch_hip_old.so python3 profiles/dp_bench.py | tail -2; KVM_LIB=$PWD/profiles/libkvmatch_hip_old.so python3 profiles/dp_lone.py
  echo "== new"; python3 profiles/dp_bench.py | tail -2; python3 profiles/dp_lone.py
done
