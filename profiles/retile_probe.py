#!/usr/bin/env python3
"""kvm_retile_cnsm_kernel alone: the bench list's cNSM-ED call under rocprofv3 is heavy; this times the CALL with the pruner
made trivial by a filter nothing passes (alpha just above 1, beta tiny): what is left is plan + retile + launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 100_000_000, 8192
s = datagen.gen_series(n)
q = s[n // 3: n // 3 + m].copy() * 1e-3 + 12345.0     # a query no window resembles: nothing passes (alpha, beta)
left, right = datagen.gen_intervals(n, m, 1_000_000, 64.0, seed=1)
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
ctx.set_option(kvmatch.OPT_TIMING, 1)
eng = kvmatch.NormQueryEngine(ctx)
for _ in range(2000): r = eng.verify((left, right), 1, q, 10.0, 1.0001, 1e-9)
ks = []
t0 = time.perf_counter()
for _ in range(200):
    r = eng.verify((left, right), 1, q, 10.0, 1.0001, 1e-9); ks.append(r.stats[kvmatch.STAT_KERNEL_NS] / 1e3)
dt = (time.perf_counter() - t0) / 200
print(f"pass filter {r.stats[kvmatch.STAT_PASS_FILTER]}  call {dt * 1e6:.1f} us  retile + pruner (events) {np.median(ks):.1f} us")
