#!/bin/bash
# same-box A/B of DP-kernel build variants: profiles/variants/*.so against the current build (dp_bench.py, dp_lone.py)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  echo "== current"; python3 profiles/dp_bench.py | tail -2; python3 profiles/dp_lone.py
  for v in "$@"; do echo "== $v"; KVM_LIB=$PWD/profiles/variants/$v python3 profiles/dp_bench.py | tail -2; KVM_LIB=$PWD/profiles/variants/$v python3 profiles/dp_lone.py; done
done
