#!/bin/bash
# Round 4 evidence, one box: (1) the default bench line; (2) rocprofv3 --kernel-trace --stats of the same command (the
# per-kernel average durations the roofline blocks must agree with: the children it starts are traced with it); (3) the same
# with one query in flight and no child runs (durations of single launches); (4) PMC passes, one --pmc group per pass and
# never together with tracing: the ED pruner over the bench with one query in flight (FETCH_SIZE / WRITE_SIZE = its HBM
# traffic; VALU / LDS counters), the packed DP kernel over dp_bench.py (16 384 full DPs).  Outputs under gpurun_out/prof_r04/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r04
rm -rf $OUT; mkdir -p $OUT
timeout 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
# (the default command starts child runs for configs[2] and [3]; under the profiler each workload is traced on its own -- and
# driven by Python threads: with eight NATIVE host threads launching at once rocprofv3's kernel tracing itself segfaults
# inside hipLaunchKernel now and then (two of three attempts; never without the profiler))
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --cpu-baseline 0 --sub-workloads 0 --python-threads 1 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cnsm_ed -- python3 bench.py --workload cnsm_ed --steps 5 --warmup 1 --extras 0 --cpu-baseline 0 --python-threads 1 > $OUT/bench_trace_cnsm_ed.json 2> $OUT/bench_trace_cnsm_ed.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cnsm_dtw -- python3 bench.py --workload cnsm_dtw --steps 5 --warmup 1 --extras 0 --cpu-baseline 0 --queries 4 --python-threads 1 > $OUT/bench_trace_cnsm_dtw.json 2> $OUT/bench_trace_cnsm_dtw.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 bench.py --cpu-baseline 0 --in-flight 1 --steps 2 --sub-workloads 0 > $OUT/bench_trace1.json 2> $OUT/bench_trace1.err
for t in trace trace1 trace_cnsm_ed trace_cnsm_dtw; do n=0; for f in $(find $OUT/$t -name "*kernel_stats.csv" | sort); do cp "$f" $OUT/${t}_kernel_stats_$n.csv; n=$((n+1)); done; done
B="python3 bench.py --steps 2 --warmup 1 --queries-per-step 20 --extras 0 --cpu-baseline 0 --in-flight 1 --sub-workloads 0"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ed/sq -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/pmc_ed/lds -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_ed/fetch -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_ed/write -- $B > /dev/null 2>&1
python3 profiles/summarize.py $OUT/pmc_ed > $OUT/pmc_ed.json
D="python3 profiles/dp_bench.py 3"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_dtw/sq -- $D > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/pmc_dtw/lds -- $D > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pmc_dtw/trace -- $D > /dev/null 2>&1
python3 profiles/summarize.py $OUT/pmc_dtw > $OUT/pmc_dtw.json
python3 - <<P
import json
for f in ("pmc_ed", "pmc_dtw"):
    d = json.load(open("$OUT/%s.json" % f))
    for g, v in d.items():
        if isinstance(v, dict):
            for k, c in v.items():
                if "ed_dot" in k or "dtw_dp" in k: print(f, g, k[:44], {a: round(b) for a, b in c.items()})
        else:
            for r in v:
                if "ed_dot" in r["kernel"] or "dtw_dp" in r["kernel"]: print(f, g, r)
P
tail -c 600 $OUT/bench_default.json
