#!/bin/bash
# the default command under rocprofv3 with native host threads, each thread's first call serialised: five attempts
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r05b; rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3 4 5; do
  KVM_BENCH_PROFILED_NATIVE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$i -- python3 bench.py --cpu-baseline 0 --steps 6 --warmup 2 > $OUT/bench_trace_$i.json 2> $OUT/bench_trace_$i.err
  echo "attempt $i: rc $?"; grep -c "SIGSEGV" $OUT/bench_trace_$i.err
  rm -rf $OUT/trace_$i
done
