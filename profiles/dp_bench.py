"""Throughput of the DP kernel alone: 16 384 full DPs (m = 4096, rho = 204; eps so large that nothing is dropped in front of
the DP or abandoned in it; DP_COUNT in the environment: another number of DPs; DP_EPS: a small eps -- say 1 -- makes
every candidate but the query itself a rejected one, which the sweep still evaluates to the end with abandoning off), KVM_OPT_TIMING = HIP-event time of the DP kernel.  dp_bench.py [dp_form ...]
dp_form = KVM_OPT_DP_KERNEL: 0 auto, 2 one candidate per wave (round 3's kernel), 3 packed form."""
import sys, time
sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 10_000_000, 4096
import os
forms = [int(a) for a in sys.argv[1:]] or [0]
NDP = int(os.environ.get("DP_COUNT", "16384"))
EPS = float(os.environ.get("DP_EPS", "1e6"))
series = datagen.gen_series(n, seed=4242)
qoff = n // 3 + 5
q = series[qoff-1:qoff-1+m].copy()
L, R = datagen.gen_intervals(n, m, NDP - 1, 64.0, must_cover=[qoff], seed=3)
ctx = kvmatch.Phase2Context(0); ctx.load_series(series)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0); ctx.set_option(kvmatch.OPT_LB_CASCADE, 0); ctx.set_option(kvmatch.OPT_TIMING, 1)
eng = kvmatch.QueryEngineDtw(ctx)
rho = 204
cells = m*(2*rho+1) - rho*(rho+1)
for form in forms:
    ctx.set_option(kvmatch.OPT_DP_KERNEL, form)
    for i in range(4):
        a = eng.verify((L, R), 1, q, EPS, 0.05)
        ns = ctx._lib.kvm_last_kernel_ns(ctx._h)
        print("form", form, "windows", a.stats[0], "answers", len(a), "dp kernel ms %.4f" % (ns*1e-6), "cells/s %.3e" % (a.stats[0]*cells/(ns*1e-9)),
              "frac of 39.3 Tops/s %.3f" % (a.stats[0]*cells*5/(ns*1e-9)/39.3e12), flush=True)
