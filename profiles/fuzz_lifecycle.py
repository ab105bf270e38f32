#!/usr/bin/env python3
"""One context through a random sequence of operations -- reloading series of other lengths, all four engines with changing
m / rho / lists, index builds, reads from HBM, envelopes, sharing contexts -- each result against the oracle: state carried
from one call to the next (scratch buffers, the cached data envelope, pinned staging) must never leak into an answer.
`fuzz_lifecycle.py [steps] [seed]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, kvmatch, oracle
from kvmatch import datagen
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle.set_lenient_sort(True)
bad = 0
def same(a, o):
    return np.array_equal(a.offsets, o[0]) and np.array_equal(a.distances, o[1]) and a.cnt_candidate == o[2]["cnt_candidate"]
ctx = kvmatch.Phase2Context(0); ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
other = None
s = None
for step in range(steps):
    op = rng.choice(["load", "verify", "verify", "verify", "verify", "index", "read", "env", "share", "option"]) if s is not None else "load"
    try:
        if op == "load":
            n = int(rng.choice([400, 3_000, 40_000, 250_000, 1_200_000]))
            s = datagen.gen_series(n, seed=int(rng.integers(1, 1 << 30))) if rng.random() < 0.7 else rng.integers(-4, 5, n).astype(np.float64)
            if rng.random() < 0.5: ctx.load_series(s)
            else:
                import tempfile
                with tempfile.NamedTemporaryFile(suffix=".dat") as f:
                    datagen.write_series_file(f.name, s); ctx.load_series_file(f.name, n)
            other = None
        elif op == "verify":
            n = len(s); m = int(rng.choice([25, 100, 333, 1024])); m = min(m, n)
            if m < 25: continue
            c = ctx if other is None or rng.random() < 0.5 else other
            qoff = int(rng.integers(1, n - m + 2)); q = s[qoff - 1:qoff - 1 + m].copy()
            try:
                L, R = datagen.gen_intervals(n, m, int(rng.integers(1, max(2, min(5000, (n - m + 1) // 4)))), float(rng.choice([1.0, 8.0, 64.0])), must_cover=[qoff], seed=int(rng.integers(1, 1 << 30)))
            except Exception:      # (the list generator refuses shapes that do not fit the series: not what is under test)
                L, R = np.array([qoff], dtype=np.int32), np.array([min(qoff + 5, n - m + 1)], dtype=np.int32)
            eps = float(rng.choice([0.0, 1.0, 5.0])); rho = int(rng.choice([0, 2, m // 20, m // 5])); a = float(rng.choice([1.0, 1.5, 1e9])); b = float(rng.choice([0.0, 3.0, 1e9]))
            v = int(rng.integers(0, 4))
            if v == 0: ok = same(kvmatch.QueryEngine(c).verify((L, R), 1, q, eps), oracle.rsm_ed(s, q, L, R, 0, eps))
            elif v == 1: ok = same(kvmatch.NormQueryEngine(c).verify((L, R), 1, q, eps, a, b), oracle.cnsm_ed(s, q, L, R, 0, eps, a, b))
            elif v == 2: ok = same(kvmatch.QueryEngineDtw(c).verify((L, R), 1, q, eps, rho), oracle.rsm_dtw(s, q, L, R, 0, eps, rho))
            else: ok = same(kvmatch.NormQueryEngineDtw(c).verify((L, R), 1, q, eps, rho, a, b, resolved=True), oracle.cnsm_dtw(s, q, L, R, 0, eps, rho, a, b))
            if not ok: bad += 1; print(f"DIFFER step {step} verify variant {v} n={n} m={m} eps={eps} rho={rho} a={a} b={b} shared={c is other}", flush=True)
        elif op == "index":
            w = int(rng.choice([25, 50, 400]))
            if len(s) < w: continue
            if ctx.index_build(w)[0] != oracle.index_file(s, w)[0]: bad += 1; print(f"DIFFER step {step} index w={w} n={len(s)}", flush=True)
        elif op == "read":
            n = len(s); ln = int(rng.integers(1, min(n, 5000) + 1)); left = int(rng.integers(1, n - ln + 2))
            if not np.array_equal(ctx.read_time_series(left, ln), s[left - 1:left - 1 + ln]): bad += 1; print(f"DIFFER step {step} read", flush=True)
        elif op == "env":
            rho = int(rng.choice([1, 7, 60])); u, l = ctx.data_envelope(rho)
            ol, ou = oracle.lower_upper_lemire(s, rho)
            if not (np.array_equal(u, ou) and np.array_equal(l, ol)): bad += 1; print(f"DIFFER step {step} envelope rho={rho}", flush=True)
        elif op == "share":
            other = kvmatch.Phase2Context(0).share_series(ctx); other.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
        else:
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, int(rng.integers(0, 2))); ctx.set_option(kvmatch.OPT_LB_CASCADE, int(rng.integers(0, 2)))
            ctx.set_option(kvmatch.OPT_DP_KERNEL, int(rng.integers(0, 2))); ctx.set_option(kvmatch.OPT_ED_KERNEL, int(rng.choice([0, 0, 1, 3])))
    except Exception as e:
        bad += 1; print(f"EXCEPTION step {step} op {op}: {type(e).__name__} {str(e)[:120]}", flush=True)
print("steps", steps, "differences", bad)
sys.exit(1 if bad else 0)
