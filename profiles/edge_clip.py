"""Interval clipping / shift / range errors: random interval lists that stick out of [1, n-m+1] on both sides, with
lastSegment > 1 shifts -- status (IllegalArgument or not) and answers must equal the oracle's for all four variants."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, kvmatch
from kvmatch import datagen
import oracle
ctx = kvmatch.Phase2Context(0)
ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1); oracle.set_lenient_sort(True)  # compare the arithmetic also where the reference's sort throws
rng = np.random.default_rng(int(os.environ.get("SEED", "4")))
base = datagen.gen_series(300_000, seed=8)
bad = tot = errs = quirk = 0
def run(fn):
    try:
        return ("ok", fn())
    except (kvmatch.KvMatchIllegalArgument,) as e:
        return ("illegal", None)
    except kvmatch.KvMatchIndexOutOfBounds as e:   # (round 3: where the reference ends in ArrayIndexOutOfBounds)
        return ("oob", None)
    except oracle.OracleError as e:
        return ({-2: "illegal", -8: "oob"}.get(e.code, f"oracle{e.code}"), None)
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    n = int(rng.integers(2_000, 60_000)); s = base[int(rng.integers(0, 200_000)):][:n].copy()
    m = int(rng.choice([25, 50, 128, 700, 1500])); m = min(m, n)
    ctx.load_series(s)
    seg = int(rng.choice([1, 1, 2, 3, 41]))          # lastSegment: shift = (seg - 1) * 25
    k = int(rng.integers(1, 40))
    pts = np.sort(rng.integers(-200, n + 300, 2 * k))
    L = pts[0::2].astype(np.int32); R = pts[1::2].astype(np.int32)
    keep = np.concatenate(([True], L[1:] > R[:-1] + 1)); L, R = L[keep], R[keep]
    if rng.uniform() < 0.3:                            # squeeze everything to one edge
        L = (L // 50 + (0 if rng.uniform() < 0.5 else n - 60)).astype(np.int32); R = np.maximum(L, (R // 50 + L)).astype(np.int32)
        order = np.argsort(L); L, R = L[order], R[order]
        keep = np.concatenate(([True], L[1:] > R[:-1] + 1)); L, R = L[keep], R[keep]
    qoff = int(rng.integers(1, n - m + 2)); q = s[qoff - 1: qoff - 1 + m].copy()
    eps = float(rng.choice([0.0, 0.3, 3.0])) * np.sqrt(m)
    sh = (seg - 1) * 25
    pairs = [("rsm_ed", lambda: kvmatch.QueryEngine(ctx).verify((L, R), seg, q, eps), lambda: oracle.rsm_ed(s, q, L, R, sh, eps)),
             ("cnsm_ed", lambda: kvmatch.NormQueryEngine(ctx).verify((L, R), seg, q, eps, 1.5, 3.0), lambda: oracle.cnsm_ed(s, q, L, R, sh, eps, 1.5, 3.0)),
             ("rsm_dtw", lambda: kvmatch.QueryEngineDtw(ctx).verify((L, R), seg, q, eps, 5), lambda: oracle.rsm_dtw(s, q, L, R, sh, eps, 5)),
             ("cnsm_dtw", lambda: kvmatch.NormQueryEngineDtw(ctx).verify((L, R), seg, q, eps, 5, 1.5, 3.0), lambda: oracle.cnsm_dtw(s, q, L, R, sh, eps, 5, 1.5, 3.0))]
    for name, g, o in pairs:
        tot += 1
        sg, a = run(g); so, w = run(o)
        if sg != so:
            bad += 1; print("STATUS", name, case, sg, so, "n", n, "m", m, "seg", seg, L[:4], R[:4], flush=True); continue
        if sg != "ok":
            errs += 1; continue
        if not (len(a) == len(w[0]) and np.array_equal(a.offsets, w[0]) and np.array_equal(a.distances, w[1]) and a.cnt_candidate == w[2]["cnt_candidate"]):
            bad += 1; print("MISMATCH", name, case, len(a), len(w[0]), a.cnt_candidate, w[2]["cnt_candidate"], "n", n, "m", m, "seg", seg, "eps", eps, "qoff", qoff,
                            "got", a.offsets[:5], a.distances[:5], "want", w[0][:5], w[1][:5], "stats", a.stats[:6], {k_: w[2][k_] for k_ in ("pass_filter", "pass_kim", "pass_keogh", "pass_keogh2")}, flush=True)
print("checks", tot, "both-illegal", errs, "tie-quirk", quirk, "bad", bad)
sys.exit(1 if bad else 0)
