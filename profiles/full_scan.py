"""Index-free full scan (every offset a candidate: U/Ucr{Ed,Dtw}QueryExecutor.java semantics) on the GPU: parity against
the restated executors at n = NP, timings at n = N.  ED: single-chain (bit-identical) and segmented; DTW: EPOCH buffers."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kvmatch
from kvmatch import datagen
import oracle

N = int(os.environ.get("N", 100_000_000))
NP = int(os.environ.get("NP", 10_000_000))
s = datagen.gen_series(N, seed=1)
ctx = kvmatch.Phase2Context(0)
out = {}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)

def run(tag, ex, reps=2):
    ex.run()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); ex.run(); best = min(best, time.perf_counter() - t0)
    r = ex.answers
    out[tag] = {"ms": best * 1e3, "candidates": int(r.cnt_candidate), "answers": len(r), "cand_per_s": r.cnt_candidate / best,
                "stats": [int(x) for x in r.stats[:6]]}
    return r

def check(tag, r, fn):
    t0 = time.perf_counter(); wo, wd, _ = fn(); dt = time.perf_counter() - t0
    out[tag]["oracle_s"] = dt
    out[tag]["parity"] = bool(np.array_equal(r.offsets, wo) and np.array_equal(r.distances, wd))
    out[tag]["same_offsets"] = bool(sorted(r.offsets.tolist()) == sorted(wo.tolist()))

for m, eps, rho, alpha, beta in ((8192, 3.0, 409, 1.5, 2.0), (1024, 2.0, 51, 1.1, 1.0)):
    qoff = 5_000_000
    for n in (NP, N):
        sub = s[:n].copy()
        q = datagen.plant_copies(sub, qoff, m, [n // 3, n - 2 * m], [1.0, 1.0], [0.0, 0.0], 0.02, seed=m)
        ctx.load_series(sub)
        tag = f"n{n:.0e}_m{m}"
        r = run(tag + "_ucr_ed_chain", kvmatch.UcrEdQueryExecutor(ctx, n, m, q, eps, alpha, beta), reps=1)
        if n == NP:
            check(tag + "_ucr_ed_chain", r, lambda: oracle.ucr_ed(sub, q, eps, alpha, beta))
        r = run(tag + "_ucr_ed_seg50k", kvmatch.UcrEdQueryExecutor(ctx, n, m, q, eps, alpha, beta, segment=50_000))
        if n == NP:
            check(tag + "_ucr_ed_seg50k", r, lambda: oracle.ucr_ed(sub, q, eps, alpha, beta))
        r = run(tag + "_ucr_dtw", kvmatch.UcrDtwQueryExecutor(ctx, n, m, q, eps, rho, alpha, beta))
        if n == NP:
            check(tag + "_ucr_dtw", r, lambda: oracle.ucr_dtw(sub, q, eps, rho, alpha, beta))
        for k in (tag + "_ucr_ed_chain", tag + "_ucr_ed_seg50k", tag + "_ucr_dtw"):
            sys.stderr.write(k + " " + json.dumps(out[k]) + "\n")
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", "full_scan.json"), "w"), indent=1)
