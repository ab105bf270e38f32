#!/bin/bash
# A/B of DP kernel builds on one box: every profiles/variants/<name>.so given on the command line runs dp_bench.py with
# the given forms.   ab_pack.sh "<forms>" name...
cd "$(dirname "$0")/.."
forms=$1; shift
for v in "$@"; do
  echo "== $v"; KVM_LIB=$PWD/profiles/variants/$v.so timeout -k 10 200 python3 profiles/dp_bench.py $forms 2>&1 | grep -v amdgpu.ids | tail -n +2
done
