#!/bin/bash
# round 5: the 16-per-lane class A (EDD_A16) -- parity tests that go through the ED pruner, then A/B against the 8-per-lane
# build on the same box: full isolated tiles and the bench list
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py tests/test_golden.py tests/test_adversarial_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "not dtw" > gpurun_out/r05_a16_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r05_a16_tests.log; echo "tests rc $rc"
[ $rc -eq 0 ] || exit $rc
for lib in "" profiles/variants/libkvmatch_hip_a8.so; do
  echo "== lib ${lib:-default (A16)}"
  KVM_LIB=${lib:+$PWD/$lib} MS=8192 LS=64,512,65536 python3 profiles/ed_run_length.py 2>&1 | grep "run length"
  KVM_LIB=${lib:+$PWD/$lib} python3 bench.py --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 > gpurun_out/r05_a16_bench_${lib:+a8}.json 2> gpurun_out/r05_a16_bench_${lib:+a8}.err; echo "bench rc $?"
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/r05_a16_bench_${lib:+a8}.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value %.4g  kernel_ms %.4f frac %.3f list_bound_ms %s full_tiles_ms %s parity %s" % (d["value"], r["kernel_ms"], r["frac"], r.get("list_bound_ms"), r.get("full_tiles_kernel_ms"), d.get("parity_vs_oracle")))
PY
done
