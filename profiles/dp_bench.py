import sys, time
sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 10_000_000, 4096
series = datagen.gen_series(n, seed=4242)
qoff = n // 3 + 5
q = series[qoff-1:qoff-1+m].copy()
L, R = datagen.gen_intervals(n, m, 16383, 64.0, must_cover=[qoff], seed=3)  # 16 384 DPs = 8 full rounds of the 2 048 waves in flight
ctx = kvmatch.Phase2Context(0); ctx.load_series(series)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0); ctx.set_option(kvmatch.OPT_LB_CASCADE, 0); ctx.set_option(kvmatch.OPT_TIMING, 1)
eng = kvmatch.QueryEngineDtw(ctx)
rho = 204
cells = m*(2*rho+1) - rho*(rho+1)
for i in range(4):
    a = eng.verify((L, R), 1, q, 1e6, 0.05)  # eps so large that nothing is dropped in front of the DP or abandoned in it
    ns = ctx._lib.kvm_last_kernel_ns(ctx._h)
    print("windows", a.stats[0], "dp kernel ms", ns*1e-6, "cells/s %.3e" % (a.stats[0]*cells/(ns*1e-9)), "cand/s %.3e" % (a.stats[0]/(ns*1e-9)))
