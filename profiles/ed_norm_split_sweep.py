#!/usr/bin/env python3
"""cNSM-ED (device-built tile lists: the pruner decides in the kernel whether to split every unit four ways, from the real
counts: units <= split_units) -- the threshold forced through KVM_ED_SPLIT_UNITS over list sizes and pass rates; early
abandon off and on; kernel time from HIP events (median of 8)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
n = 100_000_000
s = datagen.gen_series(n)
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
ctx.set_option(kvmatch.OPT_TIMING, 1)
eng = kvmatch.NormQueryEngine(ctx)
thr = ("0", "512", "1024", "2048", "4096", "8192", "auto")
for ab in (0, 1):
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
    for m in (8192, 1024):
        q = s[n // 3: n // 3 + m].copy()
        for c in (200_000, 500_000, 1_000_000, 2_000_000, 4_000_000):
            for alpha, beta in ((1e9, 1e9), (2.0, 10.0)):
                left, right = datagen.gen_intervals(n, m, c, 64.0, seed=c % 97)
                row = []
                for t in thr:
                    if t == "auto": os.environ.pop("KVM_ED_SPLIT_UNITS", None)
                    else: os.environ["KVM_ED_SPLIT_UNITS"] = t
                    for _ in range(6 if t == thr[0] else 3): r = eng.verify((left, right), 1, q, 10.0, alpha, beta)
                    ks = [eng.verify((left, right), 1, q, 10.0, alpha, beta).stats[kvmatch.STAT_KERNEL_NS] / 1e6 for _ in range(8)]
                    row.append(float(np.median(ks)))
                best = int(np.argmin(row[:-1]))
                print(f"abandon {ab} m {m:5d} cands {c:8d} alpha {alpha:g} beta {beta:g} pass {int(r.stats[kvmatch.STAT_PASS_FILTER]):8d}: " + "  ".join(f"{t} {v:.4f}" for t, v in zip(thr, row)) + f"   best {thr[best]}  auto/best {row[-1] / row[best]:.3f}", flush=True)
