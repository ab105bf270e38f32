#!/usr/bin/env python3
"""Does the ED pruner's guard band hold against COHERENT rounding?  A self-match (true distance 0) whose window is 0, A, v, v,
v, ...: the dot product's chain adds the same tiny product p = v (v - mean q) to an accumulator of ~A^2 a million times, and
every addition rounds the same way by the same fraction of an ulp -- the worst case a length-m bound must cover.  Scans v;
prints for each whether the library still returns the self-match for eps^2 = 2e-12.   guard_band_coherent.py [m]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, kvmatch
from fractions import Fraction
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
n = m + 4096
U = Fraction(1, 2**53)
lost = []
with kvmatch.Phase2Context(0) as ctx:
    for k in range(0, 60):
        v = 1e-8 * (1.0 + k * 1.37e-3)
        s = np.full(n, v)
        qoff = 1001
        s[qoff - 1] = 0.0
        s[qoff] = 1.0
        q = s[qoff - 1:qoff - 1 + m].copy()
        qbar = float(np.sum(q) / m)
        p = Fraction(v) * Fraction(v - qbar)
        frac = float(p / U - (p / U).__floor__())
        ctx.load_series(s)
        L = np.array([qoff], dtype=np.int32); R = np.array([qoff + 2], dtype=np.int32)
        res = {}
        for ab in (1, 0):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
            a = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 1.4e-6)
            res[ab] = (qoff in a.offsets.tolist(), a.stats[kvmatch.STAT_RECHECKED] if hasattr(kvmatch, "STAT_RECHECKED") else None)
        print(f"k={k:2d} v={v:.12e} frac(p/ulp)={frac:.3f} self-match kept: abandon on {res[1][0]}, off {res[0][0]}", flush=True)
        if not (res[1][0] and res[0][0]): lost.append(k)
print("LOST", lost)
