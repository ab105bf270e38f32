"""UcrEd single-chain mode at n = 1e7: time per point of the exact running-sum replay, parity with the restated executor."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kvmatch
from kvmatch import datagen
import oracle
n = int(os.environ.get("N", 10_000_000))
s = datagen.gen_series(n, seed=1)
ctx = kvmatch.Phase2Context(0)
for m in (1024, 8192):
    sub = s.copy()
    q = datagen.plant_copies(sub, n // 2, m, [n // 3, n - 2 * m], [1.0, 1.0], [0.0, 0.0], 0.02, seed=m)
    ctx.load_series(sub)
    ex = kvmatch.UcrEdQueryExecutor(ctx, n, m, q, 2.0, 1.1, 1.0)
    ex.run()
    t0 = time.perf_counter(); ex.run(); dt = time.perf_counter() - t0
    wo, wd, _ = oracle.ucr_ed(sub, q, 2.0, 1.1, 1.0)
    ok = np.array_equal(ex.answers.offsets, wo) and np.array_equal(ex.answers.distances, wd)
    print(f"m={m}: {dt*1e3:.1f} ms, {dt/n*1e9:.2f} ns/point, answers {len(wo)}, parity {ok}", flush=True)
