#!/bin/bash
# round 5, first GPU call: (1) the A16 micro-benchmark beside the round-3 loop variants and the library's full-tile rate on the
# same box; (2) the GPU suite on the build with the ADVICE r04 fixes; (3) the default bench under rocprofv3 with NATIVE host
# threads (persistent worker pool), three attempts
set -o pipefail
mkdir -p gpurun_out
cd profiles/micro && ./ed_a16 > ../../gpurun_out/r05_ed_a16.log 2>&1; echo "ed_a16 rc $?"
./ed_loop_variants > ../../gpurun_out/r05_ed_loop_variants.log 2>&1; echo "ed_loop_variants rc $?"
cd ../..
MS=8192 LS=64,512,65536 python3 profiles/ed_run_length.py > gpurun_out/r05_ed_run_length.log 2>&1; echo "ed_run_length rc $?"
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_tests_1.log 2>&1; rc=$?; echo "gpu tests rc $rc"; tail -3 gpurun_out/r05_gpu_tests_1.log
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  KVM_BENCH_PROFILED_NATIVE=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05_prof_native_$i -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --sub-workloads 0 --extras 0 --cpu-baseline 0 > $GRAFT_REPO_ROOT/gpurun_out/r05_prof_native_$i.log 2>&1
  echo "profiled native attempt $i rc $?"
done
