#!/usr/bin/env python3
"""Random malformed interval lists (overlaps, nesting, repeats, empties, out-of-range ends, shifts, optionally unsorted)
through all four engines, library vs oracle.  `fuzz_lists.py [cases] [seed] [kmax] [wide]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, kvmatch, oracle
from kvmatch import datagen
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 12
wide = len(sys.argv) > 4 and sys.argv[4] == "wide"   # long queries and wide bands (both DP kernels), short lists
n = 60_000
s = datagen.gen_series(n, seed=21)
oracle.set_lenient_sort(True)
bad = exc = total = 0
with kvmatch.Phase2Context(0) as ctx:
    ctx.load_series(s); ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
    if os.environ.get("FUZZ_ABANDON") == "0": ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)   # host-planned tile classes A / B / C + window splits
    hostile = len(sys.argv) > 4 and sys.argv[4] == "hostile"   # integer / piecewise-constant / NaN-sprinkled / huge-level series
    for case in range(cases):
        if hostile and case % 10 == 0:
            kind = (case // 10) % 5
            if kind == 0: s = rng.integers(-3, 4, n).astype(np.float64)
            elif kind == 1: s = np.repeat(rng.normal(size=n // 50 + 1), 50)[:n].copy()
            elif kind == 2: s = datagen.gen_series(n, seed=case + 1); s[rng.integers(0, n, 40)] = np.nan; s[rng.integers(0, n, 10)] = np.inf
            elif kind == 3: s = 1e7 + np.cumsum(rng.normal(size=n))
            else: s = rng.integers(-9, 10, n) * 0.5
            ctx.load_series(s)
        m = int(rng.choice([700, 1600, 2500])) if wide else int(rng.choice([25, 64, 200, 700]))
        qoff = int(rng.integers(1, n - m))
        q = s[qoff - 1:qoff - 1 + m].copy()
        k = int(rng.integers(0, kmax))
        centres = rng.integers(-50, n + 50, k)
        if rng.random() < 0.7 and k:
            centres[rng.integers(0, k)] = qoff            # usually cover the self match
        L = centres - rng.integers(0, 40, k)
        R = centres + rng.integers(-5, 60, k)              # some right < left
        if rng.random() < 0.75:
            o = np.argsort(L, kind="stable"); L, R = L[o], R[o]
        L, R = L.astype(np.int32), R.astype(np.int32)
        ls = int(rng.choice([1, 1, 2, 3]))
        shift = (ls - 1) * 25
        eps = float(rng.choice([0.0, 0.5, 3.0, 50.0])) if not hostile else float(rng.choice([0.0, 1.0, 2.0, np.sqrt(m), 8.0])); rho = int(rng.choice([m // 10, m // 3, m // 2, m, 3 * m])) if wide else int(rng.choice([0, 3, m // 10])); a = float(rng.choice([1.0, 1.5, 1e9])); b = float(rng.choice([0.0, 2.0, 1e9]))
        for vname, lib, orc in (
            ("rsm_ed", lambda: kvmatch.QueryEngine(ctx).verify((L, R), ls, q, eps), lambda: oracle.rsm_ed(s, q, L, R, shift, eps)),
            ("cnsm_ed", lambda: kvmatch.NormQueryEngine(ctx).verify((L, R), ls, q, eps, a, b), lambda: oracle.cnsm_ed(s, q, L, R, shift, eps, a, b)),
            ("rsm_dtw", lambda: kvmatch.QueryEngineDtw(ctx).verify((L, R), ls, q, eps, rho), lambda: oracle.rsm_dtw(s, q, L, R, shift, eps, rho)),
            ("cnsm_dtw", lambda: kvmatch.NormQueryEngineDtw(ctx).verify((L, R), ls, q, eps, rho, a, b, resolved=True), lambda: oracle.cnsm_dtw(s, q, L, R, shift, eps, rho, a, b))):
            total += 1
            try:
                r = lib(); ares = sorted(zip(r.distances.tolist(), r.offsets.tolist())), r.cnt_candidate
            except kvmatch.KvMatchIllegalArgument as e:
                ares = "EXC illegal argument"
            except kvmatch.KvMatchIndexOutOfBounds as e:   # (round 3: where the reference ends in ArrayIndexOutOfBounds)
                ares = "EXC index out of bounds"
            try:
                o = orc(); ores = sorted(zip(o[1].tolist(), o[0].tolist())), o[2]["cnt_candidate"]
            except oracle.OracleError as e:
                ores = {-2: "EXC illegal argument", -8: "EXC index out of bounds"}.get(e.code, f"EXC {e.code}")
            if isinstance(ares, str) or isinstance(ores, str):
                exc += 1
            if ares != ores:
                bad += 1
                from collections import Counter
                if not isinstance(ares, str) and not isinstance(ores, str):
                    ca, co = Counter(ares[0]), Counter(ores[0])
                    only_lib, only_orc = list((ca - co).elements())[:6], list((co - ca).elements())[:6]
                    detail = f"answers {len(ares[0])} vs {len(ores[0])} cnt {ares[1]} vs {ores[1]}; only lib {only_lib}; only oracle {only_orc}"
                else:
                    detail = f"lib {str(ares)[:80]} orc {str(ores)[:80]}"
                print(f"DIFFER case {case} {vname} m={m} ls={ls} eps={eps} rho={rho} a={a} b={b} k={k} sorted={bool(np.all(np.diff(L) >= 0))}: {detail}", flush=True)
                if os.environ.get("FUZZ_DUMP"):
                    np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_case_{case}_{vname}.npz"), L=L, R=R, q=q, m=m, ls=ls, eps=eps, rho=rho, a=a, b=b)
print("engine calls", total, "with an exception on either side", exc, "differences", bad)
sys.exit(1 if bad else 0)
