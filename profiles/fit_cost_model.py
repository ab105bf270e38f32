#!/usr/bin/env python3
"""Re-fit the reference's phase-2 cost model for the GPU path.

The reference stops phase 1 early when its estimate of phase-2 time exceeds the time another index probe would save
(K/QueryEngine.java:316-327), using  T_2[ms] ~= A * #disjoint_intervals + B * (#offsets / 1e5) * m  with constants fitted
to the JAVA phase 2 (A = 4.0708, B = 0.26983, K/QueryEngine.java:55-57; A = 9.7228, B = 0.010674 for the other three
engines).  With phase 2 on an MI355X these constants are off by orders of magnitude, so phase 1 should stop far
earlier.  This script measures T_2 of the library (reference behaviour: early abandon on) over a grid of
(#intervals, #offsets, m) for the four variants and fits  T_2[ms] = C + A * #intervals + B * (#offsets/1e5) * m.
Output: JSON on stdout (committed as profiles/r01_phase2_cost_model.json)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen

n = 20_000_000
series = datagen.gen_series(n, seed=77)
ctx = kvmatch.Phase2Context(0); ctx.load_series(series)
rows = {"rsm_ed": [], "cnsm_ed": [], "rsm_dtw": [], "cnsm_dtw": []}
for m in (1024, 4096, 8192):
    qoff = n // 3
    q = series[qoff - 1: qoff - 1 + m].copy()
    for offsets in (10_000, 100_000, 1_000_000, 4_000_000):
        for mean_len in (4.0, 64.0, 1024.0):
            L, R = datagen.gen_intervals(n, m, offsets, mean_len, must_cover=[qoff], seed=5)
            for name, eng, args in (("rsm_ed", kvmatch.QueryEngine(ctx), (10.0,)),
                                    ("cnsm_ed", kvmatch.NormQueryEngine(ctx), (10.0, 1.5, 5.0)),
                                    ("rsm_dtw", kvmatch.QueryEngineDtw(ctx), (10.0, 0.05)),
                                    ("cnsm_dtw", kvmatch.NormQueryEngineDtw(ctx), (10.0, 0.05, 1.5, 5.0))):
                if name.endswith("dtw") and (m == 8192 or offsets > 1_000_000):
                    continue
                eng.verify((L, R), 1, q, *args)
                ts = []
                for _ in range(3):
                    t = time.perf_counter(); eng.verify((L, R), 1, q, *args); ts.append(time.perf_counter() - t)
                rows[name].append((len(L), offsets, m, min(ts) * 1e3))
out = {"model": "T_2[ms] = C + A * intervals + B * (offsets / 1e5) * m", "n": n,
       "reference_java_constants": {"rsm_ed": {"A": 4.0708, "B": 0.26983}, "others": {"A": 9.7228, "B": 0.010674}}}
for name, r in rows.items():
    r = np.array(r)
    X = np.stack([np.ones(len(r)), r[:, 0], r[:, 1] / 1e5 * r[:, 2]], axis=1)
    coef, *_ = np.linalg.lstsq(X, r[:, 3], rcond=None)
    pred = X @ coef
    out[name] = {"C_ms": coef[0], "A_ms_per_interval": coef[1], "B": coef[2], "samples": len(r),
                 "max_abs_err_ms": float(np.abs(pred - r[:, 3]).max()), "t2_ms_range": [float(r[:, 3].min()), float(r[:, 3].max())]}
print(json.dumps(out, indent=1))
