import sys, os
sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 10_000_000, 8192
series = datagen.gen_series(n)
qoff = n // 3 + 17
q = series[qoff - 1:qoff - 1 + m].copy()
ctx = kvmatch.Phase2Context(0); ctx.load_series(series)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
eng = kvmatch.NormQueryEngine(ctx)
for before in (0, 350, 700, 2800):
    L = np.array([qoff - before], dtype=np.int32); R = np.array([qoff], dtype=np.int32)
    for i in range(3):
        a = eng.verify((L, R), 1, q, 1.0, 2.0, 10.0)
    print("before", before, "answers", len(a), file=sys.stderr)
