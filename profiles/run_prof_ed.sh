cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rocprofv3 -L > gpurun_out/prof/counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python3 bench.py --steps 5 --warmup 1 --extras 0 --cpu-baseline 0 > gpurun_out/prof/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/pmc1 -- python3 bench.py --steps 2 --warmup 0 --extras 0 --cpu-baseline 0 > gpurun_out/prof/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/prof/pmc2 -- python3 bench.py --steps 2 --warmup 0 --extras 0 --cpu-baseline 0 > gpurun_out/prof/pmc2.log 2>&1
ls -R gpurun_out/prof | head -40
