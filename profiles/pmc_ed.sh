# PMC passes over the ED pruners on a uniform workload (64-candidate runs x 65536): what are the waves waiting for?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_ed
mkdir -p $OUT
for K in ${KERNELS:-3 4}; do
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/k${K}_sq -- python3 profiles/pmc_ed.py $K 64 65536 > /dev/null 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/k${K}_lds -- python3 profiles/pmc_ed.py $K 64 65536 > /dev/null 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH --output-format csv -d $OUT/k${K}_misc -- python3 profiles/pmc_ed.py $K 64 65536 > /dev/null 2>&1
done
python3 profiles/summarize.py $OUT
