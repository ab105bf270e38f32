#!/bin/bash
# kernel trace of the latency-bound streams with F queries in flight, and how their kernels overlap.  prof_stream.sh F rsm|cnsm
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F=$1; W=$2
rm -rf gpurun_out/prof_stream_$W$F; mkdir -p gpurun_out/prof_stream_$W$F
STREAM_ONLY=$W STREAM_CALLS=300 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_stream_$W$F -- python3 profiles/stream_scaling.py $F 2>&1 | grep "in flight"
python3 profiles/overlap.py gpurun_out/prof_stream_$W$F 0.3
