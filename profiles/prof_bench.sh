#!/bin/bash
# rocprofv3 summary of the bench command itself (kernel trace + stats), then the HBM-traffic counter passes on the same
# command, each in its own run.  Outputs: gpurun_out/prof_bench/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_bench
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 > $OUT/bench_write.json 2> $OUT/bench_write.err
find $OUT -name "*kernel_stats.csv" | head -2
true
