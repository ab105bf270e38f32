#!/bin/bash
# Round 3, the evidence of the final build on one box, in this order: the default bench line and rocprofv3 kernel stats of
# the same command (prof_r03.sh); the DP kernel's same-box A/B against the build before the round's change and its PMC
# passes (ab_dtw.sh, pmc_dtw.sh); whole DTW calls under rocprofv3 (prof_dtw_calls.sh); the cost-model shapes (dtw_shapes.py).
cd "$GRAFT_REPO_ROOT"
bash profiles/prof_r03.sh > gpurun_out/r03_final_bench.log 2>&1
timeout -k 10 300 bash profiles/ab_dtw.sh > gpurun_out/r03_ab_dtw.log 2>&1
bash profiles/pmc_dtw.sh old KVM_LIB=$PWD/profiles/libkvmatch_hip_old.so > gpurun_out/r03_pmc_dtw_old.log 2>&1
bash profiles/pmc_dtw.sh new > gpurun_out/r03_pmc_dtw_new.log 2>&1
bash profiles/prof_dtw_calls.sh > gpurun_out/r03_prof_dtw_calls.log 2>&1
timeout -k 10 300 python3 profiles/dtw_shapes.py > gpurun_out/r03_dtw_shapes_final.log 2>&1
KVM_LIB=$PWD/profiles/libkvmatch_hip_old.so timeout -k 10 300 python3 profiles/dtw_shapes.py > gpurun_out/r03_dtw_shapes_old.log 2>&1
tail -2 gpurun_out/r03_final_bench.log | cut -c1-400
