#!/bin/bash
# Round 6 evidence, one box, in this order.  Everything DESIGN.md section 0 quotes comes out of this script + profiles/stamp_r06.py.
#  (1) the default bench line (configs[1] headline, configs[2] and [3] as streams, group_world1);
#  (2) rocprofv3 --kernel-trace --stats of the SAME command, and of the headline with one query in flight (single launches);
#  (3) PMC passes, one --pmc group per pass and never together with tracing, the program directly behind `--`:
#      the RSM-ED pruner (FETCH_SIZE / WRITE_SIZE = its HBM traffic; VALU / LDS counters), the cNSM-ED pruner + its retile kernel,
#      the banded DP on the 66 762-DP slice (LDS bank conflicts, occupancy, VALU);
#  (4) the launched path with one rank (RCCL communicator of one), the group layer's stream probe, the enqueue lock's cost.
# Outputs: gpurun_out/prof_r06/ ; then `python3 profiles/stamp_r06.py gpurun_out/prof_r06` files the summaries under profiles/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r06
STEP=${1:-all}
if [ $STEP = all ]; then rm -rf $OUT; fi
mkdir -p $OUT
if [ $STEP = all ] || [ $STEP = bench ]; then
timeout -k 10 1000 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc $?"
fi
if [ $STEP = all ] || [ $STEP = trace ]; then
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 bench.py --cpu-baseline 0 > $OUT/bench_trace.json 2> $OUT/bench_trace.err; echo "profiled default command rc $?"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_if1 -- python3 bench.py --cpu-baseline 0 --in-flight 1 --steps 2 --sub-workloads 0 > $OUT/bench_trace_if1.json 2> $OUT/bench_trace_if1.err; echo "in-flight 1 rc $?"
for t in trace_default trace_if1; do n=0; for f in $(find $OUT/$t -name "*kernel_stats.csv" | sort); do cp "$f" $OUT/${t}_kernel_stats_$n.csv; n=$((n+1)); done; done
fi
if [ $STEP = all ] || [ $STEP = pmc ]; then
B="python3 bench.py --steps 2 --warmup 1 --queries-per-step 20 --extras 0 --cpu-baseline 0 --in-flight 1 --sub-workloads 0"
for W in rsm_ed cnsm_ed; do
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_$W/sq -- $B --workload $W > /dev/null 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/pmc_$W/lds -- $B --workload $W > /dev/null 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$W/fetch -- $B --workload $W > /dev/null 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_$W/write -- $B --workload $W > /dev/null 2>&1
  python3 profiles/summarize.py $OUT/pmc_$W > $OUT/pmc_$W.json; echo "pmc $W done"
done
timeout -k 10 300 python3 profiles/dp_slice.py 3 > $OUT/dp_slice.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_dtw/sq -- python3 profiles/dp_slice.py 1 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/pmc_dtw/lds -- python3 profiles/dp_slice.py 1 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dtw -- python3 profiles/dp_slice.py 3 > /dev/null 2>&1
for f in $(find $OUT/trace_dtw -name "*kernel_stats.csv" | sort); do cp "$f" $OUT/trace_dtw_kernel_stats.csv; done
python3 profiles/summarize.py $OUT/pmc_dtw > $OUT/pmc_dtw.json; echo "pmc dtw done"
fi
if [ $STEP = all ] || [ $STEP = group ]; then
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --extras 0 --cpu-baseline 0 --steps 8 > $OUT/bench_launched_world1.json 2> $OUT/bench_launched_world1.err; echo "launched world 1 rc $?"
timeout -k 10 300 python3 profiles/tools/group_stream_probe.py > $OUT/group_stream_probe.log 2>&1
for e in 0 1; do echo "== KVM_SERIALIZE_ENQUEUE=$e"; KVM_SERIALIZE_ENQUEUE=$e timeout -k 10 200 python3 profiles/tools/group_stream_probe.py --only host --depths 8 2>&1 | grep path; done > $OUT/enqueue_lock_cost.log
fi
rm -rf $OUT/trace_default $OUT/trace_if1 $OUT/trace_dtw $OUT/pmc_*/*/*/*.db 2>/dev/null
du -sh $OUT; tail -c 300 $OUT/bench_default.json 2>/dev/null
