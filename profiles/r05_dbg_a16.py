#!/usr/bin/env python3
"""Debug: the long list of short runs (m = 300) of tests/test_parity_gpu.py through both forms of class A and every split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 3_000_000, int(os.environ.get("DBG_M", "300"))
s = datagen.gen_series(n, seed=99)
rng = np.random.default_rng(99)
q = s[1_000_000:1_000_000 + m].copy()
L, R, pos = [], [], 1
while pos + 48 + m < n:
    ln = int(rng.integers(1, 49)); L.append(pos); R.append(pos + ln - 1); pos += ln + int(rng.integers(2, 13))
L, R = np.array(L, dtype=np.int32), np.array(R, dtype=np.int32)
planted = []
for i in rng.choice(len(L), 300, replace=False):
    off = int(rng.integers(L[i], R[i] + 1))
    if planted and min(abs(off - p) for p in planted) < 2 * m: continue
    s[off - 1: off - 1 + m] = q + rng.normal(size=m) * 0.01; planted.append(off)
if os.environ.get("DBG_UPPER"):
    keep = L > 1_500_000
    L, R = L[keep].copy(), R[keep].copy()
    planted = [p for p in planted if p > 1_500_000]
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
eng = kvmatch.QueryEngine(ctx)
ref = None
lens = R - L + 1
for form in ("0", "1"):
    for sp in ("",):
        if form: os.environ["KVM_ED_A16"] = form
        else: os.environ.pop("KVM_ED_A16", None)
        if sp: os.environ["KVM_ED_SPLIT"] = sp
        else: os.environ.pop("KVM_ED_SPLIT", None)
        os.environ["KVM_TRACE"] = "1"
        got = eng.verify((L, R), 1, q, 2.0)
        os.environ.pop("KVM_TRACE")
        offs = set(got.offsets.tolist())
        ol = got.offsets.tolist()
        dups = sorted(o for o in offs if ol.count(o) > 1)
        if dups:
            di = np.searchsorted(L, dups, side="right") - 1
            a_runs = np.flatnonzero(lens >= 33)
            print("duplicated answers:", [(o, ol.count(o), int(lens[i]), int(np.searchsorted(a_runs, i))) for o, i in zip(dups, di)][:40])
        extra = sorted(offs - set(planted))
        print("answers that were not planted:", extra[:20], "dist", [float(got.distances[ol.index(o)]) for o in extra[:5]])
        if ref is None: ref = offs
        if ref is not None and ref - offs:
            miss = sorted(ref - offs)
            idx = np.searchsorted(L, miss, side="right") - 1
            a_runs = np.flatnonzero(lens >= 33)
            rk = np.searchsorted(a_runs, idx)
            print("missing (rank, WG, wave, row, lane class):", [(int(r), int(r) // 64, (int(r) // 16) % 4, int(r) % 16, int(o - L[i]) // 16) for r, o, i in zip(rk, miss, idx)][:60])
            print("missing: run lengths", lens[idx][:40].tolist(), "\npos in run", (np.array(miss) - L[idx])[:40].tolist(), "\nrank among runs >= 33:", np.searchsorted(a_runs, idx)[:40].tolist(), "of", len(a_runs))
            fnd = sorted(ref & offs); idf = np.searchsorted(L, fnd, side="right") - 1
            big = lens[idf] >= 33
            print("found in runs >= 33 (rank, WG, wave, row, lane class):", [(int(r), int(r) // 64, (int(r) // 16) % 4, int(r) % 16, int(o - L[i]) // 16) for r, o, i in zip(np.searchsorted(a_runs, idf[big]), np.array(fnd)[big], idf[big])][:120])
        print(f"form {form or 'auto'} split {sp or 'auto'}: {len(got)} answers, planted found {len(set(planted) & offs)}/{len(planted)}, missing vs first {sorted(ref - offs)[:6]}", flush=True)
