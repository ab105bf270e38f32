#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_host
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_host -- python3 profiles/trace_host.py > gpurun_out/prof_host.log 2>&1
f=$(find gpurun_out/prof_host -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | head -12 | cut -c1-160
