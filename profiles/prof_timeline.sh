#!/bin/bash
# GPU timeline (kernels + copies) of the last RSM-ED call of profiles/trace_host.py
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_tl; mkdir -p gpurun_out/prof_tl
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_tl -- python3 profiles/trace_host.py > gpurun_out/prof_tl.log 2>&1
python3 profiles/timeline.py gpurun_out/prof_tl
grep elapsed_ms gpurun_out/prof_tl.log | tail -2
