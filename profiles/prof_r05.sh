#!/bin/bash
# Round 5 evidence, one box: (1) the default bench line; (2) rocprofv3 --kernel-trace --stats of the SAME command, three
# attempts, its contexts driven by the library's NATIVE host threads (the persistent worker pool of round 5: round 4's
# fresh-threads-per-call form made the profiler's library fall over two times in three); (3) the headline with one query in
# flight (durations of single launches); (4) PMC passes over the ED pruner with one query in flight, one --pmc group per pass
# and never together with tracing (FETCH_SIZE / WRITE_SIZE = its HBM traffic; VALU / LDS counters).  Outputs: gpurun_out/prof_r05/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r05
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc $?"
for i in 1 2 3; do
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$i -- python3 bench.py --cpu-baseline 0 > $OUT/bench_trace_$i.json 2> $OUT/bench_trace_$i.err
  echo "profiled default command, native threads, attempt $i: rc $?"
done
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 bench.py --cpu-baseline 0 --in-flight 1 --steps 2 --sub-workloads 0 > $OUT/bench_trace1.json 2> $OUT/bench_trace1.err; echo "in-flight 1 rc $?"
for t in trace_1 trace_2 trace_3 trace1; do n=0; for f in $(find $OUT/$t -name "*kernel_stats.csv" | sort); do cp "$f" $OUT/${t}_kernel_stats_$n.csv; n=$((n+1)); done; done
B="python3 bench.py --steps 2 --warmup 1 --queries-per-step 20 --extras 0 --cpu-baseline 0 --in-flight 1 --sub-workloads 0"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ed/sq -- $B > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/pmc_ed/lds -- $B > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_ed/fetch -- $B > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_ed/write -- $B > /dev/null 2>&1
python3 profiles/summarize.py $OUT/pmc_ed > $OUT/pmc_ed.json
python3 - <<P
import json
d = json.load(open("$OUT/pmc_ed.json"))
for g, v in d.items():
    if isinstance(v, dict):
        for k, c in v.items():
            if "ed_dot" in k: print("pmc_ed", g, k[:44], {a: round(b) for a, b in c.items()})
    else:
        for r in v:
            if "ed_dot" in r["kernel"]: print("pmc_ed", g, r)
P
rm -rf $OUT/trace_1 $OUT/trace_2 $OUT/trace_3 $OUT/trace1 $OUT/pmc_ed/*/*/*.db 2>/dev/null
du -sh $OUT; tail -c 400 $OUT/bench_default.json
