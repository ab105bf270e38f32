"""Debug helper: the first DTW cases of the adversarial generator, one call at a time, with KVM_DEBUG_SYNC=1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kvmatch
from kvmatch import datagen
from oracle import oracle
import adversarial
rng = np.random.default_rng(11)
n = int(rng.integers(30000, 120000)); s = adversarial.make("ints", n, rng)
m = int(rng.choice([25, 31, 64, 100, 257, 1024, 2500])); rho = int(rng.choice([0, 1, max(1, m // 20), max(1, m // 10)]))
for attempt in range(50):
    qoff = int(rng.integers(1, n - m + 1)); w = s[qoff - 1: qoff - 1 + m]
    if np.all(np.isfinite(w)) and np.std(w) > 0: break
pl = [int(p) for p in rng.integers(1, n - m + 1, 3)]
q = datagen.plant_copies(s, qoff, m, pl, [1.0, 1.0, 1.0], [0.0, 0.0, 1.0], 0.0, seed=0)
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
L, R = datagen.gen_intervals(n, m, int(rng.integers(500, 8000)), float(rng.choice([1.0, 8.0, 64.0, 400.0])), must_cover=[qoff] + pl, seed=0)
print("n", n, "m", m, "rho", rho, "intervals", len(L), flush=True)
for eps in (0.0, float(np.sqrt(m)), 2.5):
    print("rsm_dtw eps", eps, flush=True)
    a = kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps, int(rho))
    o = oracle.rsm_dtw(s, q, L, R, 0, eps, rho)
    print("  got", len(a), "want", len(o[0]), "stats", a.stats, "same", np.array_equal(a.offsets, o[0]) and np.array_equal(a.distances, o[1]), flush=True)
    print("cnsm_dtw eps", eps, flush=True)
    a = kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, eps, int(rho), 1.5, 0.5)
    o = oracle.cnsm_dtw(s, q, L, R, 0, eps, rho, 1.5, 0.5)
    print("  got", len(a), "want", len(o[0]), "stats", a.stats, "same", np.array_equal(a.offsets, o[0]) and np.array_equal(a.distances, o[1]), flush=True)
