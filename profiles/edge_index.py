"""Index rows (kvm_index_rows_build) on adversarial series against the restated reference loop, bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, kvmatch
import oracle, adversarial
ctx = kvmatch.Phase2Context(0)
rng = np.random.default_rng(2)
bad = tot = 0
for kind in adversarial.KINDS + ("zeros", "negzero"):
    for n in (260_000, 100_025):
        if kind == "zeros": s = np.zeros(n)
        elif kind == "negzero": s = np.where(rng.uniform(0, 1, n) < 0.5, -0.0, 0.0) * 1.0
        else: s = adversarial.make(kind, n, rng)
        ctx.load_series(s)
        for w in (25, 50, 400):
            if w > n: continue
            k, f, l = ctx.index_rows(w)
            wk, wf, wl = oracle.index_rows(s, w)
            ok = len(k) == len(wk) and np.array_equal(f, wf) and np.array_equal(l, wl) and np.array_equal(k.view(np.int64), wk.view(np.int64))
            tot += 1
            if not ok:
                bad += 1
                print("BAD", kind, n, w, len(k), len(wk), flush=True)
                if len(k) == len(wk):
                    d = np.nonzero(k.view(np.int64) != wk.view(np.int64))[0][:3]; print("   keys differ at", d, k[d], wk[d])
print("index rows checks", tot, "bad", bad)
sys.exit(1 if bad else 0)
