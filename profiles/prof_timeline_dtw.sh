#!/bin/bash
# GPU timeline (kernels + copies) of the last selective cNSM-DTW call of profiles/trace_dtw2.py
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_tl2; mkdir -p gpurun_out/prof_tl2
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_tl2 -- python3 profiles/trace_dtw2.py > gpurun_out/prof_tl2.log 2>&1
python3 profiles/timeline.py gpurun_out/prof_tl2
grep elapsed_ms gpurun_out/prof_tl2.log | tail -3 | cut -c1-200
