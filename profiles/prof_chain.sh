#!/bin/bash
# kernel durations of the two sequential-chain kernels: index rows (n = 1e8) and the UcrEd single-chain replay (n = 1e7)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_idx $R/gpurun_out/prof_chain
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_idx -o idx -- python3 $R/profiles/index_rows.py > $R/gpurun_out/prof_idx.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_chain -o chain -- python3 $R/profiles/chain_mode.py > $R/gpurun_out/prof_chain.log 2>&1
cd $R
for f in $(find gpurun_out/prof_idx gpurun_out/prof_chain -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f | cut -c1-200; done
