#!/usr/bin/env python3
"""What the library does with interval lists that are NOT in the form sortAndMergeIntervals leaves (adjacent, overlapping,
unsorted, empty / reversed intervals), next to the oracle's restatement of the Java loops on the same lists."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, kvmatch, oracle
from kvmatch import datagen
n, m = 300_000, 200
s = datagen.gen_series(n, seed=3)
q = s[50_000:50_000 + m].copy()
for p in (20_000, 120_000, 200_000):
    s[p - 1:p - 1 + m] = q + np.random.default_rng(p).normal(size=m) * 0.01
lists = {
    "adjacent": ([100, 200, 300, 19_990, 20_001], [199, 299, 399, 20_000, 20_050]),
    "overlapping": ([100, 150, 19_990, 19_995], [200, 400, 20_010, 20_020]),
    "unsorted": ([120_000, 100, 19_990, 50_000], [120_100, 400, 20_010, 50_040]),
    "nested": ([100, 150, 160], [1000, 200, 170]),
    "reversed": ([500, 19_990], [400, 20_010]),
    "duplicate": ([19_990, 19_990, 19_990], [20_010, 20_010, 20_010]),
}
oracle.set_lenient_sort(True)
with kvmatch.Phase2Context(0) as ctx:
    ctx.load_series(s); ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
    for name, (L, R) in lists.items():
        L, R = np.array(L, dtype=np.int32), np.array(R, dtype=np.int32)
        for vname, lib, orc in (("rsm_ed", lambda: kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 3.0), lambda: oracle.rsm_ed(s, q, L, R, 0, 3.0)),
                                ("cnsm_ed", lambda: kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 1.0, 1.5, 3.0), lambda: oracle.cnsm_ed(s, q, L, R, 0, 1.0, 1.5, 3.0)),
                                ("rsm_dtw", lambda: kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, 3.0, 10), lambda: oracle.rsm_dtw(s, q, L, R, 0, 3.0, 10)),
                                ("cnsm_dtw", lambda: kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 1.0, 10, 1.5, 3.0), lambda: oracle.cnsm_dtw(s, q, L, R, 0, 1.0, 10, 1.5, 3.0))):
            try:
                a = lib(); ares = (a.offsets.tolist(), a.distances.tolist(), a.cnt_candidate)
            except Exception as e:
                ares = ("EXC", type(e).__name__, str(e)[:60])
            try:
                o = orc(); ores = (o[0].tolist(), o[1].tolist(), o[2]["cnt_candidate"])
            except Exception as e:
                ores = ("EXC", type(e).__name__, str(e)[:60])
            same = ares == ores or (ares[0] == "EXC" and ores[0] == "EXC")
            print(f"{name:12s} {vname:9s} {'same' if same else 'DIFFER'}  lib {str(ares)[:90]}  | oracle {str(ores)[:90]}", flush=True)
