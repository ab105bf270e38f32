#!/usr/bin/env python3
"""Uniform ED workload for PMC runs: `pmc_ed.py KERNEL LENGTH COUNT` (RSM-ED, abandon off, m=8192, 3 launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
k, length, count = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n, m = 100_000_000, 8192
series = datagen.gen_series(n)
q = series[n // 3:n // 3 + m].copy()
ctx = kvmatch.Phase2Context(0); ctx.load_series(series)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0); ctx.set_option(kvmatch.OPT_ED_KERNEL, k)
stride = (n - m - length) // count
L = (1 + np.arange(count, dtype=np.int64) * stride).astype(np.int32); R = (L + length - 1).astype(np.int32)
eng = kvmatch.QueryEngine(ctx)
for _ in range(3):
    a = eng.verify((L, R), 1, q, 10.0)
print("windows", a.stats[0])
