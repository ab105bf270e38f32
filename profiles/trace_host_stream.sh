#!/bin/bash
# kernel trace (not only stats) of the host-path streams: who occupies the GPU in the early-abandoning stream?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_host; rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$i -- python3 bench.py --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 --sub-workloads 0 > $OUT/bench$i.json 2> $OUT/bench$i.err
  rc=$?; echo "attempt $i rc $rc"
  if [ $rc = 0 ]; then cp $(find $OUT/t$i -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv; break; fi
done
ls -la $OUT | head; rm -rf $OUT/t1 $OUT/t2 $OUT/t3
