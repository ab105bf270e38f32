# PMC passes over the ED pruners on a uniform workload (64-candidate runs x 65536): what are the waves waiting for?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_ed
mkdir -p $OUT
for K in 3 4; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/k${K}_sq -- python3 profiles/pmc_ed.py $K 64 65536 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/k${K}_lds -- python3 profiles/pmc_ed.py $K 64 65536 > /dev/null 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum TD_BUSY_avr --output-format csv -d $OUT/k${K}_tcp -- python3 profiles/pmc_ed.py $K 64 65536 > /dev/null 2>&1
done
python3 profiles/summarize.py $OUT
