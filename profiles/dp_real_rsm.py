"""The DP kernel on real RSM-DTW queries (no normalisation): the bench's mixed-stream leg (m = 8192, rho = 409, eps = 10) and
the same at m = 4096 -- candidates near the query reach the DP, early abandoning off / on.  dp_real_rsm.py m eps [dp_form ...]"""
import sys, time, os
sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n = 100_000_000
m = int(sys.argv[1]); eps = float(sys.argv[2])
forms = [int(a) for a in sys.argv[3:]] or [0]
series = datagen.gen_series(n, seed=0x4B564D31)
qoff = n // 3 + 17
q = series[qoff - 1: qoff - 1 + m].copy()
L, R = datagen.gen_intervals(n, m, int(float(os.environ.get("DP_CANDIDATES", "1e6"))), 64.0, must_cover=[qoff], seed=2)
ctx = kvmatch.Phase2Context(0); ctx.load_series(series)
ctx.set_option(kvmatch.OPT_TIMING, 1)
eng = kvmatch.QueryEngineDtw(ctx)
rho = int(ctx._lib.kvm_resolve_rho(0.05, m))
cells = m * (2 * rho + 1) - rho * (rho + 1)
ref = None
for ab in (0, 1):
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
    for form in forms:
        ctx.set_option(kvmatch.OPT_DP_KERNEL, form)
        for i in range(4):
            t = time.perf_counter(); a = eng.verify((L, R), 1, q, eps, 0.05); dt = time.perf_counter() - t
            ns = ctx._lib.kvm_last_kernel_ns(ctx._h)
            key = (a.offsets.tobytes(), a.distances.tobytes())
            if ref is None: ref = key
            if i >= 2:
                print("m", m, "abandon", ab, "form", form, "dp", a.stats[4], "answers", len(a), "same", key == ref, "dp kernel ms %.4f" % (ns * 1e-6),
                      "call ms %.3f" % (dt * 1e3), "frac(full) %.3f" % (a.stats[4] * cells * 5 / (ns * 1e-9) / 39.3e12), flush=True)
