import sys, os
sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 100_000_000, 4096
series = datagen.gen_series(n)
qoff = n // 3 + 17
planted = [n // 7, n // 2 + 1000, n - 3 * m]
q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.02, -0.03], 0.01)
L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff] + planted, seed=1)
ctx = kvmatch.Phase2Context(0); ctx.load_series(series)
for eps, alpha, beta in ((1.0, 2.0, 10.0), (1.0, 1.1, 1.0), (1.0, 1e9, 1e9)):
    eng = kvmatch.NormQueryEngineDtw(ctx)
    for i in range(3):
        a = eng.verify((L, R), 1, q, eps, 204, alpha, beta)
    print("eps", eps, "alpha", alpha, "beta", beta, "elapsed_ms", a.elapsed_ms, "answers", len(a), "stats", a.stats, file=sys.stderr)
