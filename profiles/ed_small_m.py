#!/usr/bin/env python3
"""RSM-ED / cNSM-ED pruner throughput for shorter queries (1e6 candidates in phase-1-shaped intervals, abandon off/on)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
n = 100_000_000
series = datagen.gen_series(n)
ctx = kvmatch.Phase2Context(0); ctx.load_series(series); ctx.set_option(kvmatch.OPT_TIMING, 1)
for m in (128, 256, 512, 1024, 2048, 4096, 8192):
    qoff = n // 3 + 17
    q = series[qoff - 1:qoff - 1 + m].copy()
    L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff], seed=1)
    for ab in (0, 1):
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
        for name, eng, args in (("rsm", kvmatch.QueryEngine(ctx), (0.3 * np.sqrt(m),)), ("cnsm", kvmatch.NormQueryEngine(ctx), (0.3 * np.sqrt(m), 2.0, 10.0))):
            eng.verify((L, R), 1, q, *args)
            t = time.perf_counter(); reps = 5
            for _ in range(reps): a = eng.verify((L, R), 1, q, *args)
            dt = (time.perf_counter() - t) / reps
            k = ctx._lib.kvm_last_kernel_ns(ctx._h) * 1e-6
            print(f"m={m:5d} abandon={ab} {name:4s}: {dt*1e3:7.3f} ms/call kernels {k:6.3f} ms  {a.cnt_candidate/dt:.3e} cand/s  "
                  f"{a.cnt_candidate*m*2/(k*1e-3)/1e12 if ab==0 and name=='rsm' else 0:5.1f} TFLOP/s answers {len(a)}")
