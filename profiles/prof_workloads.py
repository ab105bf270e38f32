#!/usr/bin/env python3
"""Small fixed workloads for rocprofv3 (run under `rocprofv3 ... -- python3 profiles/prof_workloads.py MODE`).

MODE ed     : the bench.py headline workload (RSM-ED, n=1e8, m=8192, 1e6 candidates, early abandon off), 3 launches
MODE calib  : FETCH_SIZE calibration -- wave-per-candidate ED kernel on ISOLATED candidates spaced 2m apart, so
              every launch must fetch exactly count*8*m bytes from HBM with the kernels' 8 B/lane access shape
MODE dtw    : DTW DP kernel alone (cascade off, early abandon off), m=4096, rho=204, 8192 candidates
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np  # noqa: E402
import kvmatch  # noqa: E402
from kvmatch import datagen  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "ed"
n = 100_000_000
series = datagen.gen_series(n)
ctx = kvmatch.Phase2Context(0)
ctx.load_series(series)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
if mode == "ed":
    m = 8192
    qoff = n // 3 + 17
    q = series[qoff - 1:qoff - 1 + m].copy()
    L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff], seed=1)
    eng = kvmatch.QueryEngine(ctx)
    for _ in range(3):
        a = eng.verify((L, R), 1, q, 10.0)
    print("ed windows", a.stats[0], "answers", len(a))
elif mode == "cnsm":
    m = 8192
    qoff = n // 3 + 17
    planted = [n // 7, n // 2 + 1000, n - 3 * m]
    q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.02, -0.03], 0.01)
    ctx.load_series(series)
    L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff] + planted, seed=1)
    eng = kvmatch.NormQueryEngine(ctx)
    for _ in range(3):
        a = eng.verify((L, R), 1, q, 10.0, 2.0, 10.0)
    print("cnsm windows", a.stats[0], "answers", len(a), "rechecked", a.stats[6])
elif mode == "calib":
    m = 8192
    q = series[999:999 + m].copy()
    starts = np.arange(1, n - m, 2 * m, dtype=np.int32)  # isolated windows, no overlap at all
    ctx.set_option(kvmatch.OPT_ED_KERNEL, 1)
    eng = kvmatch.QueryEngine(ctx)
    for _ in range(3):
        a = eng.verify((starts, starts), 1, q, 1.0)
    print("calib windows", a.stats[0], "bytes_per_launch", int(a.stats[0]) * 8 * m)
elif mode == "dtw":
    m = 4096
    qoff = n // 3 + 17
    q = series[qoff - 1:qoff - 1 + m].copy()
    L, R = datagen.gen_intervals(n, m, 8192, 64.0, must_cover=[qoff], seed=2)
    ctx.set_option(kvmatch.OPT_LB_CASCADE, 0)
    eng = kvmatch.NormQueryEngineDtw(ctx)
    for _ in range(3):
        a = eng.verify((L, R), 1, q, 1e6, 0.05, 1e9, 1e9)  # (eps so large that nothing stops in front of or inside the DP)
    rho = 204
    print("dtw windows", a.stats[0], "cells_per_launch", int(a.stats[0]) * (m * (2 * rho + 1) - rho * (rho + 1)))
ctx.close()
