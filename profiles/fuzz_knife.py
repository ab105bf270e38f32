#!/usr/bin/env python3
"""Thresholds placed ON a candidate's decision quantity, fuzzed: for a random series / query / list, take the oracle's
wide-open answer list, pick candidates from it and query again with epsilon at that candidate's distance, one ulp to
either side, and 1e-15 .. 1e-11 relatively off (inside every pruner's guard band, so the decision falls to the exact
pass); for the cNSM engines also with beta at a candidate's |mean - meanQ| and alpha at its sigma ratio (same offsets).
All four engines, library vs oracle.  `fuzz_knife.py [cases] [seed] [kmax]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, kvmatch, oracle
from kvmatch import datagen
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 6
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
n = 40_000
oracle.set_lenient_sort(True)
bad = total = flips = 0
OFFS = (0.0, 1e-15, -1e-15, 1e-13, -1e-13, 1e-11, -1e-11)


def around(x):
    out = [x, float(np.nextafter(x, 0.0)), float(np.nextafter(x, np.inf))]
    out += [x * (1.0 + o) for o in OFFS[1:]]
    return [v for v in out if v >= 0.0 and np.isfinite(v)]


with kvmatch.Phase2Context(0) as ctx:
    ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
    if os.environ.get("FUZZ_ABANDON") == "0": ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)   # host-planned tile classes A / B / C + window splits
    for case in range(cases):
        kind = int(rng.integers(0, 5))
        if kind == 0: s = datagen.gen_series(n, seed=1000 * seed + case)
        elif kind == 1: s = rng.integers(-4, 5, n).astype(np.float64)
        elif kind == 2: s = np.repeat(rng.normal(size=n // 20 + 1), 20)[:n].copy()
        elif kind == 3: s = 1e6 + np.cumsum(rng.normal(size=n))
        else: s = np.round(np.cumsum(rng.normal(size=n)) * 4.0) / 4.0
        ctx.load_series(s)
        m = int(rng.choice([16, 25, 64, 200, 500]))
        qoff = int(rng.integers(1, n - m))
        q = s[qoff - 1:qoff - 1 + m].copy()
        mode = int(rng.integers(0, 3))
        if mode == 1: q = q + rng.normal(size=m) * 0.05 * (np.std(q) + 1e-3)
        elif mode == 2: q = np.round(q + rng.integers(-1, 2, m))
        k = int(rng.integers(1, kmax))
        centres = np.sort(rng.integers(1, n - m, k)); centres[int(rng.integers(0, k))] = qoff; centres = np.sort(centres)
        L = np.maximum(1, centres - rng.integers(0, 60, k)).astype(np.int32)
        R = (centres + rng.integers(0, 80, k)).astype(np.int32)
        ls = int(rng.choice([1, 1, 2])); shift = (ls - 1) * 25
        rho = int(rng.choice([0, 2, max(1, m // 10)]))
        big = 1e30
        engines = (
            ("rsm_ed", lambda e, a, b: kvmatch.QueryEngine(ctx).verify((L, R), ls, q, e), lambda e, a, b: oracle.rsm_ed(s, q, L, R, shift, e)),
            ("cnsm_ed", lambda e, a, b: kvmatch.NormQueryEngine(ctx).verify((L, R), ls, q, e, a, b), lambda e, a, b: oracle.cnsm_ed(s, q, L, R, shift, e, a, b)),
            ("rsm_dtw", lambda e, a, b: kvmatch.QueryEngineDtw(ctx).verify((L, R), ls, q, e, rho), lambda e, a, b: oracle.rsm_dtw(s, q, L, R, shift, e, rho)),
            ("cnsm_dtw", lambda e, a, b: kvmatch.NormQueryEngineDtw(ctx).verify((L, R), ls, q, e, rho, a, b, resolved=True), lambda e, a, b: oracle.cnsm_dtw(s, q, L, R, shift, e, rho, a, b)))
        for vname, lib, orc in engines:
            norm = vname.startswith("cnsm")
            try:
                o = orc(big if not norm else 1e150, 1e9, 1e9)
            except oracle.OracleError:
                continue
            offs, dists = o[0], o[1]
            if len(offs) == 0:
                continue
            picks = rng.choice(len(offs), size=min(3, len(offs)), replace=False)
            trials = []
            for p in picks:
                d = float(dists[p])
                if np.isfinite(d):
                    trials += [(e, 1e9, 1e9) for e in around(d)]
                if norm:
                    w = s[int(offs[p]) - 1:int(offs[p]) - 1 + m]
                    bq = abs(float(np.mean(w)) - float(np.mean(q)))
                    sw, sq = float(np.std(w)), float(np.std(q))
                    if sw > 0 and sq > 0:
                        ra = max(sw / sq, sq / sw)
                        trials += [(d * 1.5 + 1.0, a, 1e9) for a in around(ra) if a >= 1.0]
                    trials += [(d * 1.5 + 1.0, 1e9, b) for b in around(bq)]
            sizes = set()
            for e, a, b in trials:
                total += 1
                try:
                    r = lib(e, a, b); ares = (sorted(zip(r.distances.tolist(), r.offsets.tolist())), r.cnt_candidate)
                except (kvmatch.KvMatchIllegalArgument, kvmatch.KvMatchIndexOutOfBounds):
                    ares = "EXC"
                try:
                    oo = orc(e, a, b); ores = (sorted(zip(oo[1].tolist(), oo[0].tolist())), oo[2]["cnt_candidate"])
                except oracle.OracleError:
                    ores = "EXC"
                if ores != "EXC": sizes.add(len(ores[0]))
                if ares != ores:
                    bad += 1
                    na = len(ares[0]) if ares != "EXC" else -1; no = len(ores[0]) if ores != "EXC" else -1
                    print(f"DIFFER case {case} kind {kind} mode {mode} {vname} m={m} ls={ls} rho={rho} eps={e!r} a={a!r} b={b!r}: answers {na} vs {no}", flush=True)
                    if os.environ.get("FUZZ_DUMP"):
                        np.savez(os.path.join(ROOT, "gpurun_out", f"knife_case_{seed}_{case}_{vname}.npz"), s=s, L=L, R=R, q=q, m=m, ls=ls, eps=e, rho=rho, a=a, b=b)
            flips += len(sizes) > 1
        if case % 5 == 4:
            print(f"  case {case + 1}: engine calls {total} differences {bad}", flush=True)
print("engine calls", total, "engine-cases in which the answer count changed across the thresholds", flips, "differences", bad)
sys.exit(1 if bad else 0)
