#!/usr/bin/env python3
"""Re-fit the reference's phase-2 cost model for the GPU path, round 2.

The reference stops phase 1 early when its estimate of phase-2 time exceeds the time another index probe would save
(K/QueryEngine.java:316-327), using  T_2[ms] ~= A * #disjoint_intervals + B * (#offsets / 1e5) * m  with constants fitted
to the JAVA phase 2 (A = 4.0708, B = 0.26983, K/QueryEngine.java:55-57; A = 9.7228, B = 0.010674 for the other three
engines).  On an MI355X the cost has a different shape: a call cannot be shorter than its launch chain (C) nor than the one
unit of work that holds an answer -- a query drawn from the series always has its self match -- and that unit walks the
whole window: D * m for the ED engines (sequential exact pass over m points), D * m^2 for the DTW engines (one banded DP
with rho = 5 % of m runs m macro-steps of work proportional to m).  So

    T_2[ms] = C + A * #intervals + (B * m + E) * (#offsets / 1e5) + D * (m | m^2 / 1e3)

(E, new in round 3: the part of a candidate's cost that does not grow with m -- its place in the tile table, its decision,
its share of a staging pass; without it the short queries of the grid, m = 1024 over 4e6 offsets, were under-estimated by
up to 39 %.)

fitted per variant on n = 1e8 with the reference's behaviour (early abandon on), by non-negative least squares on
RELATIVE residuals (the estimate feeds a ratio test, so a 20 % miss matters equally at 0.2 ms and at 5 ms).
Output: JSON on stdout (committed as profiles/r04_phase2_cost_model.json) and, with OUT_INC=<path>, the constants as a C
initialiser (kv-match_amd/csrc/kvm_cost_model.inc, which kvm_phase2_time_estimate_ms includes: one source for the numbers;
tests/test_full_size_gpu.py::test_phase2_time_model_against_measured_calls holds the estimate to measured calls)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
from scipy.optimize import nnls

n = int(os.environ.get("N", 100_000_000))
ROWS_IN = os.environ.get("FIT_ROWS")   # re-fit from the measured points of an earlier run (its JSON), no GPU needed
series = None if ROWS_IN else datagen.gen_series(n, seed=77)
ctx = None if ROWS_IN else kvmatch.Phase2Context(0)
if ctx: ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)   # timing only: a query JDK's TimSort would throw on costs what any other costs
rows = {"rsm_ed": [], "cnsm_ed": [], "rsm_dtw": [], "cnsm_dtw": []}
if ROWS_IN:
    rows = {k: v["points_intervals_offsets_m_ms_answers"] for k, v in json.load(open(ROWS_IN)).items() if isinstance(v, dict) and "samples" in v}
for m in (() if ROWS_IN else (1024, 2048, 4096, 8192)):
    qoff = n // 3
    planted = [n // 7, n // 2 + 1000]
    s2 = series.copy() if m == 1024 else s2
    q = datagen.plant_copies(s2, qoff, m, planted, [1.0, 1.0], [0.0, 0.02], 0.01)
    ctx.load_series(s2)
    for offsets in (10_000, 100_000, 1_000_000, 4_000_000):
        for mean_len in (4.0, 64.0, 1024.0):
            L, R = datagen.gen_intervals(n, m, offsets, mean_len, must_cover=[qoff] + planted, seed=5)
            # selective queries, the regime the index serves: eps scaled with sqrt(m) so that the answers stay the planted ones
            # (the time of an answer-heavy query also depends on its number of answers, which no a-priori model can know)
            e = 10.0 * (m / 8192.0) ** 0.5
            for name, eng, args in (("rsm_ed", kvmatch.QueryEngine(ctx), (e,)),
                                    ("cnsm_ed", kvmatch.NormQueryEngine(ctx), (e, 1.5, 5.0)),
                                    ("rsm_dtw", kvmatch.QueryEngineDtw(ctx), (e, 0.05)),
                                    ("cnsm_dtw", kvmatch.NormQueryEngineDtw(ctx), (e, 0.05, 1.5, 5.0))):
                if name.endswith("dtw") and (m == 8192 or offsets > 1_000_000):
                    continue
                a = eng.verify((L, R), 1, q, *args)
                ts = []
                for _ in range(5):
                    t = time.perf_counter(); a = eng.verify((L, R), 1, q, *args); ts.append(time.perf_counter() - t)
                rows[name].append((len(L), offsets, m, float(np.median(ts)) * 1e3, len(a)))
out = {"model": "T_2[ms] = C + A * intervals + (B * m + E) * (offsets / 1e5) + D * (m for ED | m^2 / 1e3 for DTW, rho = 5 % of m)", "n": n,
       "fit": "non-negative least squares on relative residuals; median of 5 calls per point; early abandon on (reference behaviour)",
       "reference_java_constants": {"rsm_ed": {"A": 4.0708, "B": 0.26983}, "others": {"A": 9.7228, "B": 0.010674}}}
for name, r in rows.items():
    r = np.array(r, dtype=np.float64)
    extra = r[:, 2] if name.endswith("ed") else r[:, 2] ** 2 / 1e3
    X = np.stack([np.ones(len(r)), r[:, 0], r[:, 1] / 1e5 * r[:, 2], extra, r[:, 1] / 1e5], axis=1)
    y = r[:, 3]
    coef, _ = nnls(X / y[:, None], np.ones(len(r)))
    pred = X @ coef
    rel = np.abs(pred - y) / pred
    out[name] = {"C_ms": coef[0], "A_ms_per_interval": coef[1], "B": coef[2], "D": coef[3], "E": coef[4], "samples": len(r),
                 "max_rel_err_of_prediction": float(rel.max()), "median_rel_err": float(np.median(rel)),
                 "max_abs_err_ms": float(np.abs(pred - y).max()), "t2_ms_range": [float(y.min()), float(y.max())],
                 "max_answers": int(r[:, 4].max()),
                 "worst_point_intervals_offsets_m_ms_answers_pred": [float(v) for v in r[int(rel.argmax())]] + [float(pred[int(rel.argmax())])],
                 "points_intervals_offsets_m_ms_answers": r.tolist()}
print(json.dumps(out, indent=1))
if os.environ.get("OUT_INC"):
    with open(os.environ["OUT_INC"], "w") as f:
        f.write("// generated by profiles/fit_cost_model.py (see profiles/r04_phase2_cost_model.json): {C [ms], A [ms per interval],\n"
                "// B [ms per 1e5 offset-points], D [ms per point (ED) | per m^2 / 1e3 (DTW)], E [ms per 1e5 offsets]} for rsm_ed, cnsm_ed, rsm_dtw, cnsm_dtw\n")
        for name in ("rsm_ed", "cnsm_ed", "rsm_dtw", "cnsm_dtw"):
            o = out[name]
            f.write("{%.6g, %.6g, %.6g, %.6g, %.6g},  // %s: median error %.0f %%, worst point %.0f %%\n" %
                    (o["C_ms"], o["A_ms_per_interval"], o["B"], o["D"], o["E"], name, 100 * o["median_rel_err"], 100 * o["max_rel_err_of_prediction"]))
