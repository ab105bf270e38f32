#!/bin/bash
# PMC passes of kvm_ed_dot_kernel on pure class-A / class-B / class-C lists (profiles/pure_class_lists.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for W in A B C; do
  OUT=gpurun_out/pmc_class_$W; rm -rf $OUT; mkdir -p $OUT
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 profiles/pure_class_lists.py $W > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/lds -- python3 profiles/pure_class_lists.py $W > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 profiles/pure_class_lists.py $W > /dev/null 2>&1
  python3 profiles/summarize.py $OUT > gpurun_out/pmc_class_$W.json
  python3 - <<P
import json
d=json.load(open("gpurun_out/pmc_class_$W.json"))
for g,v in d.items():
    if isinstance(v,dict):
        for k,c in v.items():
            if 'ed_dot' in k: print("$W",g,{a:round(b) for a,b in c.items()})
    else:
        for r in v:
            if 'ed_dot' in r['kernel']: print("$W",g,r['calls'],round(r['avg_us'],1))
P
done
