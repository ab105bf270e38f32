import sys, os
sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 100_000_000, 8192
series = datagen.gen_series(n)
qoff = n // 3 + 17
planted = [n // 7, n // 2 + 1000, n - 3 * m]
q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.02, -0.03], 0.01)
L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff] + planted, seed=1)
ctx = kvmatch.Phase2Context(0); ctx.load_series(series)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, int(os.environ.get("AB", "0")))
eng = kvmatch.NormQueryEngine(ctx)
for i in range(4):
    print("---- call", i, file=sys.stderr)
    a = eng.verify((L, R), 1, q, 10.0, 2.0, 10.0)
    print("elapsed_ms", a.elapsed_ms, "answers", len(a), file=sys.stderr)
