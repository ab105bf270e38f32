#!/usr/bin/env python3
"""The DP kernel on the slice bench.py's `roofline_dtw.full_dp` times: cNSM-DTW, n = 1e8, m = 4096, rho = 204, the first 1024
intervals of the 1e6-candidate list (~66 762 candidates), cascade and early abandoning off and eps so large that no bound
ever reaches it -- every candidate runs ALL its cells.  Prints the kernel's HIP-event time; under `rocprofv3 --pmc ...` this
is the run whose counters profiles/stamp_r06.py files under profiles/dtw_pmc.json.   dp_slice.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, m = 100_000_000, 4096
series = datagen.gen_series(n)
qoff = n // 3 + 17
q = series[qoff - 1: qoff - 1 + m].copy()
L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff], seed=2)
L, R = L[:1024], R[:1024]
rho = 204
cells = m * (2 * rho + 1) - rho * (rho + 1)
with kvmatch.Phase2Context(0) as ctx:
    ctx.load_series(series)
    ctx.set_option(kvmatch.OPT_LB_CASCADE, 0)
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
    ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
    eng = kvmatch.NormQueryEngineDtw(ctx)
    eng.verify((L, R), 1, q, 1e6, 0.05, 1e9, 1e9)
    ctx.set_option(kvmatch.OPT_TIMING, 1)
    for _ in range(reps):
        a = eng.verify((L, R), 1, q, 1e6, 0.05, 1e9, 1e9)
        ns = ctx._lib.kvm_last_kernel_ns(ctx._h)
        print("full DPs %d  kernel %.3f ms  %.3e cells/s  frac of 39.3 Tops/s %.3f" % (
            a.stats[0], ns * 1e-6, a.stats[0] * cells / (ns * 1e-9), a.stats[0] * cells * 5 / (ns * 1e-9) / 39.3e12), flush=True)
