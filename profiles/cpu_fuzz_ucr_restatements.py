#!/usr/bin/env python3
"""CPU only: the oracle's restatements of the index-free executors (U/UcrEdQueryExecutor.java, U/UcrDtwQueryExecutor.java)
against its restatements of the engines on the matching candidate lists, on integer-valued series (exact ties), and a
classification of every difference.  The executor mirrors of the library run the ENGINES' semantics, so each class is
either reproduced by the mirror or a documented difference:
  ED : eps^2 == 0 artefact (reproduced) | executor-only answers at distance eps: a partial sum exactly on eps^2 (not reproduced)
  DTW: engine-only answers at distance eps: the executor's uncleared cb1 / cb2 at an exact LB_Keogh tie (not reproduced)
`cpu_fuzz_ucr_restatements.py [seed] [runs]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, oracle
from kvmatch.ucr import segments, EPOCH
oracle.set_lenient_sort(True)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 600
ed = dict(runs=0, zero=0, partial=0, other=0); dtw = dict(runs=0, stale=0, other=0)
for case in range(runs):
    n = int(rng.choice([2000, 6000, 20000])); m = int(rng.choice([25, 32, 60, 100]))
    kind = case % 3
    s = (rng.integers(-3, 4, n).astype(np.float64) if kind == 0 else np.round(np.cumsum(rng.normal(size=n))) * 0.5 if kind == 1
         else np.repeat(rng.integers(-2, 3, n // 4 + 1), 4)[:n].astype(np.float64))
    qoff = int(rng.integers(0, n - m)); q = s[qoff:qoff + m].copy()
    if q.std() == 0: q[0] += 1
    for p in rng.integers(0, n - m, 3): s[p:p + m] = q
    eps = float(rng.choice([0.0, 1.0, 2.0, 3.0, np.sqrt(m) / 2])); rho = int(rng.choice([0, 1, 2, m // 10]))
    a = float(rng.choice([1.0, 1.5, 1e9])); b = float(rng.choice([0.0, 1.0, 1e9]))
    uo, ud, _ = oracle.ucr_ed(s, q, eps, a, b)
    eo, ee, _ = oracle.cnsm_ed(s, q, np.array([1], dtype=np.int32), np.array([n - m + 1], dtype=np.int32), 0, eps, a, b)
    ed["runs"] += 1
    if not (np.array_equal(uo, eo) and np.array_equal(ud, ee)):
        su, se = dict(zip(uo.tolist(), ud.tolist())), dict(zip(eo.tolist(), ee.tolist()))
        if eps * eps == 0.0: ed["zero"] += 1
        elif set(se) <= set(su) and all(su[o] == se[o] for o in se) and all(su[o] == eps for o in su if o not in se): ed["partial"] += 1
        else: ed["other"] += 1; print(f"ED OTHER case {case} n={n} m={m} eps={eps} a={a} b={b}", flush=True)
    Ls, Rs = segments(1, n - m + 1, EPOCH - m + 1)
    try:
        uo, ud, _ = oracle.ucr_dtw(s, q, eps, rho, a, b); eo, ee, _ = oracle.cnsm_dtw(s, q, Ls, Rs, 0, eps, rho, a, b)
    except oracle.OracleError:
        continue
    dtw["runs"] += 1
    if not (np.array_equal(uo, eo - 1) and np.array_equal(ud, ee)):
        su, se = set(zip(uo.tolist(), ud.tolist())), set(zip((eo - 1).tolist(), ee.tolist()))
        if su < se and all(d == eps for _, d in se - su): dtw["stale"] += 1
        else: dtw["other"] += 1; print(f"DTW OTHER case {case} n={n} m={m} eps={eps} rho={rho} a={a} b={b}", flush=True)
print("ED  executor vs engine:", ed)
print("DTW executor vs engine:", dtw)
sys.exit(1 if ed["other"] or dtw["other"] else 0)
