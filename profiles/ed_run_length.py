#!/usr/bin/env python3
"""Kernel time of the ED pruner on 1e6 candidates in runs of equal length (every tile a full class-A tile): isolated tiles
(64), one unit per run (512), long runs -- the same arithmetic, different memory locality.  RSM-ED m = 8192, abandon off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
n = 100_000_000
s = datagen.gen_series(n)
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0); ctx.set_option(kvmatch.OPT_TIMING, 1)
eng = kvmatch.QueryEngine(ctx)
WARM = 3000
import os
MS = [int(x) for x in os.environ.get('MS', '8192,4096,2048').split(',')]
LS = [int(x) for x in os.environ.get('LS', '64,128,512,4096,65536').split(',')]
for m, L in [(mm, ll) for mm in MS for ll in LS]:
    q = s[n // 3: n // 3 + m].copy()
    runs = 1_000_000 // L
    starts = 1 + np.arange(runs, dtype=np.int64) * ((n - m - L - 100) // runs)
    left = starts.astype(np.int32); right = (starts + L - 1).astype(np.int32)
    for _ in range(WARM): r = eng.verify((left, right), 1, q, 10.0)   # (the first configuration also brings the clocks up: ~1 s of load)
    WARM = 5
    ks = []
    for _ in range(20):
        r = eng.verify((left, right), 1, q, 10.0); ks.append(r.stats[kvmatch.STAT_KERNEL_NS] / 1e6)
    k = float(np.median(ks)); c = runs * L
    print(f"m {m:5d} run length {L:6d} x {runs:6d} runs ({c:8d} candidates): kernel {k:.4f} ms  {c / k / 1e6:.3f} Gcand/s  {2 * m * c / k / 1e9:.1f} TFLOP/s", flush=True)
