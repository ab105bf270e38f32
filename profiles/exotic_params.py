#!/usr/bin/env python3
"""Exotic parameter values through all four engines, library vs oracle: eps^2 overflowing to inf, eps = inf, alpha / beta
inf or NaN, rho >= m, NaN / Inf inside the query, a constant query, denormal-scale data."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, kvmatch, oracle
from kvmatch import datagen
n, m = 100_000, 64
s = datagen.gen_series(n, seed=8)
q0 = s[30_000:30_000 + m].copy()
L = np.array([100, 29_990, 70_000], dtype=np.int32); R = np.array([160, 30_010, 70_030], dtype=np.int32)
oracle.set_lenient_sort(True)
inf, nan = float("inf"), float("nan")
cases = []
for eps in (1e200, inf, 1e-300, 0.0):
    cases.append(("eps=%g" % eps, q0, dict(eps=eps, alpha=1.5, beta=3.0, rho=5)))
for a, b in ((inf, 3.0), (1.5, inf), (nan, 3.0), (1.5, nan), (1e-9, 3.0), (1.5, -1.0)):
    cases.append(("alpha=%g beta=%g" % (a, b), q0, dict(eps=3.0, alpha=a, beta=b, rho=5)))
for rho in (0, m - 1, m, 10 * m):
    cases.append(("rho=%d" % rho, q0, dict(eps=3.0, alpha=1.5, beta=3.0, rho=rho)))
qn = q0.copy(); qn[10] = nan
qi = q0.copy(); qi[10] = inf
cases += [("query NaN", qn, dict(eps=3.0, alpha=1.5, beta=3.0, rho=5)), ("query Inf", qi, dict(eps=3.0, alpha=1.5, beta=3.0, rho=5)),
          ("query const", np.full(m, 2.5), dict(eps=3.0, alpha=1.5, beta=3.0, rho=5)),
          ("query huge", q0 * 1e150, dict(eps=3.0, alpha=1.5, beta=3.0, rho=5))]
bad = 0
with kvmatch.Phase2Context(0) as ctx:
    ctx.load_series(s); ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
    for name, q, p in cases:
        for vname, lib, orc in (
            ("rsm_ed", lambda: kvmatch.QueryEngine(ctx).verify((L, R), 1, q, p["eps"]), lambda: oracle.rsm_ed(s, q, L, R, 0, p["eps"])),
            ("cnsm_ed", lambda: kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, p["eps"], p["alpha"], p["beta"]), lambda: oracle.cnsm_ed(s, q, L, R, 0, p["eps"], p["alpha"], p["beta"])),
            ("rsm_dtw", lambda: kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, p["eps"], p["rho"]), lambda: oracle.rsm_dtw(s, q, L, R, 0, p["eps"], p["rho"])),
            ("cnsm_dtw", lambda: kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, p["eps"], p["rho"], p["alpha"], p["beta"], resolved=True), lambda: oracle.cnsm_dtw(s, q, L, R, 0, p["eps"], p["rho"], p["alpha"], p["beta"]))):
            try:
                a = lib(); ares = (a.offsets.tolist(), a.distances.tolist(), a.cnt_candidate)
            except Exception as e:
                ares = ("EXC", type(e).__name__, str(e)[:50])
            try:
                o = orc(); ores = (o[0].tolist(), o[1].tolist(), o[2]["cnt_candidate"])
            except Exception as e:
                ores = ("EXC", type(e).__name__, str(e)[:50])
            same = (ares[0] == "EXC" and ores[0] == "EXC") or (ares[0] != "EXC" and ores[0] != "EXC" and ares[0] == ores[0] and ares[2] == ores[2]
                    and np.array_equal(np.array(ares[1]), np.array(ores[1]), equal_nan=True))
            bad += not same
            print(f"{name:22s} {vname:9s} {'same  ' if same else 'DIFFER'} lib n={len(ares[0]) if ares[0] != 'EXC' else ares[1:]} oracle n={len(ores[0]) if ores[0] != 'EXC' else ores[1:]}", flush=True)
print("differences:", bad)
