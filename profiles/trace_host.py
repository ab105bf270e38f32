#!/usr/bin/env python3
"""KVM_TRACE host-stage timings of the bench query (RSM-ED, m = 8192, 1e6 candidates), abandon off / on; run with
KVM_TRACE=1 [KVM_LIB=...]; prints the stderr trace of the last call of each mode."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 100_000_000, 8192
s = datagen.gen_series(n)
qoff = n // 3 + 17
planted = [n // 7, n // 2 + 1000, n - 3 * m]
q = datagen.plant_copies(s, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.02, -0.03], 0.01)
L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff] + planted, seed=1)
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
eng = kvmatch.QueryEngine(ctx)
for ab in (0, 1):
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
    sys.stderr.flush()
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(2); os.dup2(devnull, 2)
    for _ in range(30): eng.verify((L, R), 1, q, 10.0)
    t = time.perf_counter()
    for _ in range(50): eng.verify((L, R), 1, q, 10.0)
    dt = (time.perf_counter() - t) / 50
    os.dup2(saved, 2)
    print(f"== abandon {ab}: {dt*1e3:.4f} ms per call", file=sys.stderr, flush=True)
    eng.verify((L, R), 1, q, 10.0)
