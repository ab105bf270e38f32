#!/bin/bash
# PMC passes over the bench (one query in flight): the lane-granular ED pruner (kvm_ed_g_kernel) and, with KVM_ED_GEN=0, round 2's
# kvm_ed_dot_kernel -- one --pmc group per pass, never together with tracing.  pmc_edg.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmc_edg_$TAG
mkdir -p $OUT
B="python3 bench.py --steps 2 --warmup 1 --queries-per-step 20 --extras 0 --cpu-baseline 0 --in-flight 1"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $B > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/lds -- $B > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > /dev/null 2>&1
python3 profiles/summarize.py $OUT > gpurun_out/pmc_edg_$TAG.json
python3 - <<P
import json
d=json.load(open("gpurun_out/pmc_edg_$TAG.json"))
for g,v in d.items():
    if isinstance(v,dict):
        for k,c in v.items():
            if 'ed_g_kernel' in k or 'ed_dot' in k: print(g,k[:30],{a:round(b) for a,b in c.items()})
    else:
        for r in v:
            if 'ed_g' in r['kernel'] or 'ed_dot' in r['kernel']: print(g,r)
P
