#!/bin/bash
# per-kernel durations of a cNSM-ED call and a cNSM-DTW call on the bench list
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_cnsm $R/gpurun_out/prof_dtw
cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cnsm -o cnsm -- python3 profiles/trace_cnsm.py > gpurun_out/prof_cnsm.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dtw -o dtw -- python3 profiles/trace_dtw.py > gpurun_out/prof_dtw.log 2>&1
grep elapsed_ms gpurun_out/prof_cnsm.log | tail -3
