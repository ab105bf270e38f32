#!/bin/bash
# Round 4, late: the register DP carrying the minima of one row in V (abandoning) / of no row (abandoning off) -- parity first
# (every DTW test, the adversarial sweep with its ties, the configs), then the DP alone and the cNSM-DTW stream.
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "dtw or adversarial or config or ucr" > gpurun_out/trk_tests.log 2>&1 || { tail -30 gpurun_out/trk_tests.log; exit 1; }
tail -2 gpurun_out/trk_tests.log
DP_COUNT=49152 timeout -k 10 200 python profiles/dp_bench.py 2 2>&1 | grep -v amdgpu.ids | tail -2 || exit 1
timeout -k 10 300 python bench.py --workload cnsm_dtw --steps 5 --warmup 1 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cnsm_dtw stream: every DP in full %.4g /s (%.3f ms per query), abandoning %.4g /s; kernel_ms %.3f frac %.3f' % (d['value'], d['ms_per_query'], d['value_early_abandon'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
