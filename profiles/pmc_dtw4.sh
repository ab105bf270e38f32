#!/bin/bash
# PMC passes over profiles/dp_bench.py (16 384 full DPs per call, m = 4096, rho = 204) for one DP form of one build -- one --pmc
# group per pass, never together with tracing; a kernel trace last.   pmc_dtw4.sh <tag> <form> [KVM_LIB=...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; FORM=$2; shift; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmc_dtw_$TAG
mkdir -p $OUT
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 profiles/dp_bench.py $FORM > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/lds -- python3 profiles/dp_bench.py $FORM > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 profiles/dp_bench.py $FORM > /dev/null 2>&1
python3 profiles/summarize.py $OUT > gpurun_out/pmc_dtw_$TAG.json
python3 - <<P
import json
d=json.load(open("gpurun_out/pmc_dtw_$TAG.json"))
for g,v in d.items():
    if isinstance(v,dict):
        for k,c in v.items():
            if 'dtw_dp' in k: print(g,k[:44],{a:round(b) for a,b in c.items()})
    else:
        for r in v:
            if 'dtw_dp' in r['kernel'] or 'prep' in r['kernel']: print(g,r)
P
