#!/usr/bin/env python3
"""CPU only: the C oracle against the Python transliteration on the ADVERSARIAL series kinds of tests/adversarial.py
(small integers, plateaus with sigma = 0, a 1e7 level, wide dynamic range, half-integers, NaN / Inf points, 1e-158 and
1e153 scales) -- where C's and Python's handling of 0/0, x/0, overflow and NaN ordering could part ways without either
noticing.  All four engines, bit for bit incl. the exception cases.  `cpu_fuzz_adversarial_restatements.py [cases] [seed]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle
import java_transliteration as J
from adversarial import KINDS, make
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 160
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle.set_lenient_sort(False)
bad = total = exc = crash = 0
for case in range(cases):
    kind = KINDS[case % len(KINDS)]
    n = int(rng.integers(1200, 3000))
    s = make(kind, n, rng)
    m = int(rng.choice([25, 31, 64]))
    rho = int(rng.choice([0, 1, max(1, m // 10)]))
    qoff = int(rng.integers(1, n - m + 1))
    for attempt in range(50):
        w = s[qoff - 1:qoff - 1 + m]
        if np.all(np.isfinite(w)) and np.std(w) > 0: break
        qoff = int(rng.integers(1, n - m + 1))
    q = s[qoff - 1:qoff - 1 + m].copy()
    pl = [int(p) for p in rng.integers(1, n - m + 1, 3)]
    for i, p in enumerate(pl): s[p - 1:p - 1 + m] = q + (1.0 if i == 2 else 0.0)
    pts = np.unique(np.concatenate([rng.integers(1, n - m + 1, 12), [qoff] + pl])).astype(np.int32)
    L = pts; R = np.minimum(pts + rng.integers(0, 30, len(pts)), n - m + 1).astype(np.int32)
    keep = np.concatenate(([True], L[1:] > R[:-1] + 1)); L, R = L[keep], R[keep]
    d0 = float(np.sum((s[pl[2] - 1:pl[2] - 1 + m] - q) ** 2))
    eps_list = [0.0, float(np.sqrt(d0)) if np.isfinite(d0) and d0 > 0 else 1.0, float(rng.uniform(0.5, 4.0))]
    a = float(rng.choice([1.0, 1.5, 1e9])); b = float(rng.choice([0.0, 2.0, 1e9]))
    sl, ql, Ll, Rl = s.tolist(), q.tolist(), [int(x) for x in L], [int(x) for x in R]
    for eps in eps_list:
        for name, py, c in (
                ("rsm_ed", lambda: J.rsm_ed(sl, ql, Ll, Rl, 1, eps), lambda: oracle.rsm_ed(s, q, L, R, 0, eps)),
                ("cnsm_ed", lambda: J.cnsm_ed(sl, ql, Ll, Rl, 1, eps, a, b), lambda: oracle.cnsm_ed(s, q, L, R, 0, eps, a, b)),
                ("rsm_dtw", lambda: J.rsm_dtw(sl, ql, Ll, Rl, 1, eps, rho), lambda: oracle.rsm_dtw(s, q, L, R, 0, eps, rho)),
                ("cnsm_dtw", lambda: J.cnsm_dtw(sl, ql, Ll, Rl, 1, eps, rho, a, b), lambda: oracle.cnsm_dtw(s, q, L, R, 0, eps, rho, a, b))):
            total += 1
            try:
                ans, cnt = py(); pres = ([x[0] for x in ans], [x[1] for x in ans], cnt)
            except J.IllegalArgumentException:
                pres = "EXC"
            except J.ArrayIndexOutOfBoundsException:
                crash += 1; continue
            try:
                o = c(); cres = (o[0].tolist(), o[1].tolist(), o[2]["cnt_candidate"])
            except oracle.OracleError:
                cres = "EXC"
            if pres == "EXC" or cres == "EXC": exc += 1
            same = pres == cres or (pres != "EXC" and cres != "EXC" and pres[0] == cres[0] and pres[2] == cres[2]
                                    and np.array_equal(np.array(pres[1]), np.array(cres[1]), equal_nan=True))
            if not same:
                bad += 1
                print(f"DIFFER case {case} {kind} {name} n={n} m={m} eps={eps} rho={rho} a={a} b={b}: {str(pres)[:100]} | {str(cres)[:100]}", flush=True)
print("engine calls", total, "with an exception on either side", exc, "reference ends in ArrayIndexOutOfBounds", crash, "differences", bad)
sys.exit(1 if bad else 0)
