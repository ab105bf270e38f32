import sys, os
sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 100_000_000, 4096
series = datagen.gen_series(n)
qoff = n // 3 + 17
q = series[qoff - 1:qoff - 1 + m].copy()
L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff], seed=1)
ctx = kvmatch.Phase2Context(0); ctx.load_series(series)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
eng = kvmatch.NormQueryEngine(ctx)
for i in range(3):
    print("---- call", i, file=sys.stderr)
    a = eng.verify((L, R), 1, q, 0.3 * np.sqrt(m), 2.0, 10.0)
    print("elapsed_ms", a.elapsed_ms, "answers", len(a), file=sys.stderr)
