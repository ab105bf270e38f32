#!/usr/bin/env python3
"""ED pruning kernels side by side on the bench workload: RSM / cNSM, early abandon on / off, kernels 2 (direct) and 3
(correlation form).  Prints ms per phase-2 call (host API) and kernel ms."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 100_000_000, 8192
series = datagen.gen_series(n)
qoff = n // 3 + 17
planted = [n // 7, n // 2 + 1000, n - 3 * m]
q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.02, -0.03], 0.01)
L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff] + planted, seed=1)
ctx = kvmatch.Phase2Context(0); ctx.load_series(series); ctx.set_option(kvmatch.OPT_TIMING, 1)
for name, eng, args in (("rsm_ed", kvmatch.QueryEngine(ctx), (10.0,)), ("cnsm_ed a=2 b=10", kvmatch.NormQueryEngine(ctx), (10.0, 2.0, 10.0)),
                        ("cnsm_ed a=1e9 b=1e9", kvmatch.NormQueryEngine(ctx), (10.0, 1e9, 1e9))):
    for ab in (0, 1):
        for k in (1, 3):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab); ctx.set_option(kvmatch.OPT_ED_KERNEL, k)
            eng.verify((L, R), 1, q, *args)
            t = time.perf_counter(); reps = 5
            for _ in range(reps): a = eng.verify((L, R), 1, q, *args)
            dt = (time.perf_counter() - t) / reps
            print(f"{name:22s} abandon={ab} kernel={k}: {dt*1e3:7.3f} ms/call  kernel {ctx._lib.kvm_last_kernel_ns(ctx._h)*1e-6:7.3f} ms  "
                  f"answers {len(a)} pass_filter {a.stats[1]} rechecked {a.stats[6]}  {a.cnt_candidate/dt:.3e} cand/s")
