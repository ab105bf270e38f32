#!/usr/bin/env python3
"""Bench workload, RSM-ED, abandon off: kernel time for the split the launcher picked (KVM_ED_SPLIT overrides)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 100_000_000, 8192
series = datagen.gen_series(n)
qoff = n // 3 + 17
q = series[qoff - 1:qoff - 1 + m].copy()
L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff], seed=1)
ctx = kvmatch.Phase2Context(0); ctx.load_series(series); ctx.set_option(kvmatch.OPT_TIMING, 1)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
eng = kvmatch.QueryEngine(ctx)
eng.verify((L, R), 1, q, 10.0)
ks = []
for _ in range(6):
    a = eng.verify((L, R), 1, q, 10.0); ks.append(ctx._lib.kvm_last_kernel_ns(ctx._h) * 1e-6)
print("split", os.environ.get("KVM_ED_SPLIT", "auto"), "kernels ms min %.3f med %.3f" % (min(ks), sorted(ks)[len(ks) // 2]))
