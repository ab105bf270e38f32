#!/usr/bin/env python3
"""Per-tile cost of the ED pruner's classes: lists of equal-length runs (64 -> one A tile, 33..63 -> one A tile partly
filled, <= 32 -> one B tile), 1e6 candidates each where possible, RSM-ED m = 8192, early abandon off; kernel time from HIP
events.  Used to price the packing of two <= 16-candidate remainders into one B row (DESIGN.md open items)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 100_000_000, 8192
s = datagen.gen_series(n)
q = s[n // 3: n // 3 + m].copy()
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0); ctx.set_option(kvmatch.OPT_TIMING, 1)
eng = kvmatch.QueryEngine(ctx)
WARM = 3000
for L, tiles in ((64, 15625), (48, 15625), (33, 15625), (32, 15625), (32, 31250), (16, 31250), (16, 62500), (8, 31250), (8, 62500), (4, 62500)):
    starts = 1 + np.arange(tiles, dtype=np.int64) * ((n - m - 100) // tiles)
    left = starts.astype(np.int32); right = (starts + L - 1).astype(np.int32)
    for _ in range(WARM): r = eng.verify((left, right), 1, q, 10.0)   # (the first list also brings the clocks up: ~1 s of load)
    WARM = 5
    ks = []
    for _ in range(20):
        r = eng.verify((left, right), 1, q, 10.0); ks.append(r.stats[kvmatch.STAT_KERNEL_NS] / 1e6)
    k = float(np.median(ks))
    print(f"run length {L:3d} x {tiles:6d} tiles ({L * tiles:8d} candidates): kernel {k:.4f} ms  = {k / tiles * 1e6:.2f} ns/tile  {L * tiles / k / 1e6:.3f} Gcand/s", flush=True)
