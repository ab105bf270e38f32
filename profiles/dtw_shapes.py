"""Whole-call time and counters of DTW calls on the shapes of the cost-model grid that a change may hurt: many offsets, few
DPs.   dtw_shapes.py"""
import sys, time
sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n = 100_000_000
series = datagen.gen_series(n, seed=77)
ctx = kvmatch.Phase2Context(0)
ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
for m in (2048, 4096):
    qoff = n // 3
    planted = [n // 7, n // 2 + 1000]
    q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.0], [0.0, 0.02], 0.01)
    ctx.load_series(series)
    e = 10.0 * (m / 8192.0) ** 0.5
    for offsets, mean_len in ((100_000, 1024.0), (1_000_000, 4.0), (1_000_000, 64.0), (1_000_000, 1024.0)):
        L, R = datagen.gen_intervals(n, m, offsets, mean_len, must_cover=[qoff] + planted, seed=5)
        for name, eng, args in (("rsm_dtw", kvmatch.QueryEngineDtw(ctx), (e, 0.05)), ("cnsm_dtw", kvmatch.NormQueryEngineDtw(ctx), (e, 0.05, 1.5, 5.0))):
            a = eng.verify((L, R), 1, q, *args)
            ts = []
            for _ in range(5):
                t = time.perf_counter(); a = eng.verify((L, R), 1, q, *args); ts.append(time.perf_counter() - t)
            st = a.stats
            print(name, "m", m, "intervals", len(L), "offsets", offsets, "ms %.3f" % (np.median(ts) * 1e3), "answers", len(a),
                  "kim", st[kvmatch.STAT_PASS_LB_KIM] if hasattr(kvmatch, "STAT_PASS_LB_KIM") else st[2], "keogh", st[3], "dp", st[4], "sequential", st[7], flush=True)
