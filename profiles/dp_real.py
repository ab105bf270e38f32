"""The DP kernel on the bench's cNSM-DTW query (BASELINE configs[3]: n = 1e8, m = 4096, rho = 204, 1e6 candidates, eps = 10,
alpha = 2, beta = 10): ~1.4e4 candidates reach the DP, a handful are answers.  Early abandoning off: every DP is evaluated
to its last cell (executed = algorithmic); on: the reference's mode.  dp_real.py [dp_form ...]"""
import sys, time, os
sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 100_000_000, 4096
forms = [int(a) for a in sys.argv[1:]] or [0]
series = datagen.gen_series(n, seed=0x4B564D31)
qoff = n // 3 + 17
q = series[qoff - 1: qoff - 1 + m].copy()
L, R = datagen.gen_intervals(n, m, int(float(os.environ.get("DP_CANDIDATES", "1e6"))), 64.0, must_cover=[qoff], seed=2)
ctx = kvmatch.Phase2Context(0); ctx.load_series(series)
ctx.set_option(kvmatch.OPT_TIMING, 1)
eng = kvmatch.NormQueryEngineDtw(ctx)
rho = 204
cells = m * (2 * rho + 1) - rho * (rho + 1)
ref = None
for ab in (0, 1):
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
    for form in forms:
        ctx.set_option(kvmatch.OPT_DP_KERNEL, form)
        for i in range(5):
            t = time.perf_counter(); a = eng.verify((L, R), 1, q, 10.0, 0.05, 2.0, 10.0); dt = time.perf_counter() - t
            ns = ctx._lib.kvm_last_kernel_ns(ctx._h)
            key = (a.offsets.tobytes(), a.distances.tobytes())
            if ref is None: ref = key
            print("abandon", ab, "form", form, "dp", a.stats[4], "answers", len(a), "same", key == ref, "dp kernel ms %.4f" % (ns * 1e-6),
                  "call ms %.3f" % (dt * 1e3), "cells/s(full) %.3e" % (a.stats[4] * cells / (ns * 1e-9)),
                  "frac %.3f" % (a.stats[4] * cells * 5 / (ns * 1e-9) / 39.3e12), flush=True)
