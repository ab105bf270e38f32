#!/bin/bash
# rocprofv3 recipe for this repo (run on the GPU box through gpurun): kernel trace + stats in one pass, every
# PMC group in its own pass (never combined with tracing).  Outputs land in gpurun_out/prof/<tag>/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof
mkdir -p $OUT
for mode in ed calib dtw; do
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${mode}_trace -- python3 profiles/prof_workloads.py $mode > $OUT/${mode}_trace.log 2>&1
  timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${mode}_fetch -- python3 profiles/prof_workloads.py $mode > $OUT/${mode}_fetch.log 2>&1
  timeout 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${mode}_write -- python3 profiles/prof_workloads.py $mode > $OUT/${mode}_write.log 2>&1
done
for mode in ed dtw; do
  timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${mode}_sq -- python3 profiles/prof_workloads.py $mode > $OUT/${mode}_sq.log 2>&1
  timeout 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/${mode}_lds -- python3 profiles/prof_workloads.py $mode > $OUT/${mode}_lds.log 2>&1
  timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${mode}_l2 -- python3 profiles/prof_workloads.py $mode > $OUT/${mode}_l2.log 2>&1
done
true
