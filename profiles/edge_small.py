"""Tiny query lengths and concurrent contexts against the oracle."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, kvmatch
from kvmatch import datagen
import oracle
bad = 0
def same(tag, a, o):
    global bad
    ok = len(a) == len(o[0]) and np.array_equal(a.offsets, o[0]) and np.array_equal(a.distances, o[1]) and a.cnt_candidate == o[2]["cnt_candidate"]
    if not ok:
        print("BAD " + tag, len(a), len(o[0]), flush=True); bad += 1
base = datagen.gen_series(400_000, seed=9)
ctx = kvmatch.Phase2Context(0)
ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1); oracle.set_lenient_sort(True)  # compare the arithmetic also where the reference's sort throws
ctx.set_option(kvmatch.OPT_STRICT_BOUNDS, 0); oracle.set_strict_bounds(False)   # ... and where it ends in ArrayIndexOutOfBounds (rho >= m below)
s = base[:50_000].copy(); ctx.load_series(s)
rng = np.random.default_rng(3)
cnt = 0
for m in (1, 2, 3, 5, 6, 7, 8, 13, 16, 24, 25):
    qoff = 20_000
    q = s[qoff - 1: qoff - 1 + m].copy()
    L, R = datagen.gen_intervals(len(s), m, 20_000, 30.0, must_cover=[qoff], seed=m)
    for eps in (0.0, 0.05 * np.sqrt(m), 0.5 * np.sqrt(m)):
        same(f"rsm_ed m={m} eps={eps:.3f}", kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps), oracle.rsm_ed(s, q, L, R, 0, eps)); cnt += 1
        if m >= 2 and np.std(q) > 0:
            same(f"cnsm_ed m={m} eps={eps:.3f}", kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, 2.0, 5.0), oracle.cnsm_ed(s, q, L, R, 0, eps, 2.0, 5.0)); cnt += 1
        if m >= 6:
            for rho in (0, 1, 2, m // 2, m - 1, m + 5):
                same(f"rsm_dtw m={m} rho={rho} eps={eps:.3f}", kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho if rho > 1 else 2), oracle.rsm_dtw(s, q, L, R, 0, eps, rho if rho > 1 else 2)); cnt += 1
                same(f"cnsm_dtw m={m} rho={rho} eps={eps:.3f}", kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho, 2.0, 5.0, resolved=True), oracle.cnsm_dtw(s, q, L, R, 0, eps, rho, 2.0, 5.0)); cnt += 1
print("tiny m:", cnt, "verifications, bad", bad, flush=True)

# two contexts used concurrently from two threads (ctypes drops the GIL during the calls)
def worker(tid, out):
    c = kvmatch.Phase2Context(0)
    c.set_option(kvmatch.OPT_SORT_VIOLATION, 1); oracle.set_lenient_sort(True)  # compare the arithmetic also where the reference's sort throws
    c.set_option(kvmatch.OPT_STRICT_BOUNDS, 0)
    ss = base[tid * 100_000: tid * 100_000 + 150_000].copy()
    c.load_series(ss)
    r = np.random.default_rng(tid)
    errs = 0
    for it in range(40):
        m = int(r.choice([64, 500, 2048]))
        qoff = int(r.integers(1, len(ss) - m))
        q = ss[qoff - 1: qoff - 1 + m].copy()
        L, R = datagen.gen_intervals(len(ss), m, 20_000, 64.0, must_cover=[qoff], seed=it + 100 * tid)
        eps = 0.2 * np.sqrt(m)
        for got, want in ((kvmatch.QueryEngine(c).verify((L, R), 1, q, eps), oracle.rsm_ed(ss, q, L, R, 0, eps)),
                          (kvmatch.NormQueryEngine(c).verify((L, R), 1, q, eps, 1.5, 3.0), oracle.cnsm_ed(ss, q, L, R, 0, eps, 1.5, 3.0)),
                          (kvmatch.NormQueryEngineDtw(c).verify((L[:50], R[:50]), 1, q, eps, 0.05, 1.5, 3.0), oracle.cnsm_dtw(ss, q, L[:50], R[:50], 0, eps, kvmatch._lib.load().kvm_resolve_rho(0.05, m), 1.5, 3.0))):
            if not (len(got) == len(want[0]) and np.array_equal(got.offsets, want[0]) and np.array_equal(got.distances, want[1])):
                errs += 1
    c.close()
    out[tid] = errs
out = {}
ts = [threading.Thread(target=worker, args=(i, out)) for i in range(3)]
[t.start() for t in ts]; [t.join() for t in ts]
print("concurrent contexts: errors per thread", out, flush=True)
bad += sum(out.values()) + (3 - len(out))
print("bad", bad)
sys.exit(1 if bad else 0)
