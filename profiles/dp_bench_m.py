"""dp_bench.py at another window length: full DPs at m (default 8192, rho = 5 %), forms given on the command line."""
import sys, os
sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n = 20_000_000
m = int(os.environ.get("DP_M", "8192")); NDP = int(os.environ.get("DP_COUNT", "8192"))
forms = [int(a) for a in sys.argv[1:]] or [0]
series = datagen.gen_series(n, seed=4242)
qoff = n // 3 + 5
q = series[qoff-1:qoff-1+m].copy()
L, R = datagen.gen_intervals(n, m, NDP - 1, 64.0, must_cover=[qoff], seed=3)
ctx = kvmatch.Phase2Context(0); ctx.load_series(series)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0); ctx.set_option(kvmatch.OPT_LB_CASCADE, 0); ctx.set_option(kvmatch.OPT_TIMING, 1)
eng = kvmatch.QueryEngineDtw(ctx)
rho = int(ctx._lib.kvm_resolve_rho(0.05, m))
cells = m*(2*rho+1) - rho*(rho+1)
ref = None
for form in forms:
    ctx.set_option(kvmatch.OPT_DP_KERNEL, form)
    for i in range(3):
        a = eng.verify((L, R), 1, q, 1e6, 0.05)
        ns = ctx._lib.kvm_last_kernel_ns(ctx._h)
        key = (a.offsets.tobytes(), a.distances.tobytes())
        if ref is None: ref = key
        print("m", m, "rho", rho, "form", form, "windows", a.stats[0], "same", key == ref, "dp kernel ms %.4f" % (ns*1e-6),
              "cells/s %.3e" % (a.stats[0]*cells/(ns*1e-9)), "frac %.3f" % (a.stats[0]*cells*5/(ns*1e-9)/39.3e12), flush=True)
