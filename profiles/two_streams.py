"""Experiment: queries in flight.  T host threads, each with its own context (own stream, staging and scratch) on the same
device, verify the bench workload concurrently; reports candidates/s for T = 1, 2, 3."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np
import kvmatch
from kvmatch import datagen
n, m = 100_000_000, 8192
series = datagen.gen_series(n)
qoff = n // 3 + 17
planted = [n // 7, n // 2 + 1000, n - 3 * m]
q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.02, -0.03], 0.01)
L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff] + planted, seed=1)
cands = int((R.astype(np.int64) - L + 1).sum())
ctxs = []
for t in range(3):
    c = kvmatch.Phase2Context(0); c.load_series(series); c.set_option(kvmatch.OPT_EARLY_ABANDON, int(os.environ.get("AB", "0"))); ctxs.append(c)
def work(c, reps, out, i):
    eng = kvmatch.QueryEngine(c)
    for _ in range(reps):
        a = eng.verify((L, R), 1, q, 10.0)
    out[i] = len(a)
for T in (1, 2, 3):
    reps = 200
    for c in ctxs[:T]: kvmatch.QueryEngine(c).verify((L, R), 1, q, 10.0)
    out = [None] * T
    th = [threading.Thread(target=work, args=(ctxs[i], reps, out, i)) for i in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    print(f"threads {T}: {T * reps * cands / dt:.4e} candidates/s, {dt / (T * reps) * 1e3:.4f} ms per query (throughput), answers {out}", flush=True)
