#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_ab; mkdir -p gpurun_out/prof_ab
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_ab -- python3 profiles/trace_abandon.py > gpurun_out/prof_ab.log 2>&1
python3 profiles/timeline.py gpurun_out/prof_ab | tail -12
grep elapsed gpurun_out/prof_ab.log | tail -2
