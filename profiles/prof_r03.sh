#!/bin/bash
# Round 3 evidence, one box: (1) the default bench line; (2) rocprofv3 --kernel-trace --stats of the same command (the
# per-kernel average durations the roofline blocks must agree with); (3) the same with one query in flight (durations of
# single launches).  Outputs under gpurun_out/prof_r03/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
timeout 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --cpu-baseline 0 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 bench.py --cpu-baseline 0 --in-flight 1 --steps 2 > $OUT/bench_trace1.json 2> $OUT/bench_trace1.err
for t in trace trace1; do f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/${t}_kernel_stats.csv; done
tail -1 $OUT/bench_default.json
