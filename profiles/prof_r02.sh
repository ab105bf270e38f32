#!/bin/bash
# Round-2 rocprofv3 recipe (run on the GPU box through gpurun).  Kernel trace + stats of the bench command with ONE query
# in flight (a kernel's rocprof duration is then its own: what bench.py reports as roofline.kernel_ms) and with the
# default three in flight; every PMC group in its own pass, never combined with tracing.  Outputs: gpurun_out/prof_r02/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r02
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --steps 10 --warmup 3 --extras 0 --cpu-baseline 0"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_f1 -- $B --in-flight 1 > $OUT/bench_f1.json 2> $OUT/bench_f1.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_f3 -- $B > $OUT/bench_f3.json 2> $OUT/bench_f3.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_dtw_f1 -- $B --in-flight 1 --workload cnsm_dtw > $OUT/bench_dtw_f1.json 2> $OUT/bench_dtw_f1.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bench_fetch -- $B --in-flight 1 --steps 3 --warmup 1 > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bench_write -- $B --in-flight 1 --steps 3 --warmup 1 > $OUT/bench_write.json 2> $OUT/bench_write.err
for mode in ed dtw; do
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${mode}_trace -- python3 profiles/prof_workloads.py $mode > $OUT/${mode}_trace.log 2>&1
  timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${mode}_sq -- python3 profiles/prof_workloads.py $mode > $OUT/${mode}_sq.log 2>&1
  timeout 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/${mode}_lds -- python3 profiles/prof_workloads.py $mode > $OUT/${mode}_lds.log 2>&1
  timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${mode}_fetch -- python3 profiles/prof_workloads.py $mode > $OUT/${mode}_fetch.log 2>&1
done
timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python3 profiles/prof_workloads.py calib > $OUT/calib_fetch.log 2>&1
python3 profiles/summarize.py $OUT > $OUT/summary.json
for t in bench_f1 bench_f3 bench_dtw_f1 ed_trace dtw_trace; do
  f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${t}_kernel_stats.csv
done
ls $OUT
true
