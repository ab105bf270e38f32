#!/usr/bin/env python3
"""Cost of a PART of an ED-pruner unit (one wave's share of 8 tiles): full class-A tiles, exactly one unit per wave slot
(2048 units), window split f = 1, 2, 4 forced -> f rounds of parts of chunks/f chunks each:  T = f a + b chunks.  The slope in f
is the per-part constant a (prologue + epilogue + hand-over), b the cost of a chunk of 128 points.  RSM-ED, early abandon off."""
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
L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
units = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
tiles = units * 8
FS = tuple(int(x) for x in os.environ.get('ED_PART_F', '1,2,4').split(','))
for m in (8192, 4096, 2048, 1024):
    q = s[n // 3: n // 3 + m].copy()
    starts = 1 + np.arange(tiles, dtype=np.int64) * ((n - m - 100) // tiles)
    left = starts.astype(np.int32); right = (starts + L - 1).astype(np.int32)
    row = []
    for f in FS:
        os.environ["KVM_ED_SPLIT"] = f"{f},{f}"
        for _ in range(WARM): r = eng.verify((left, right), 1, q, 10.0)
        WARM = 5
        ks = []
        for _ in range(20):
            r = eng.verify((left, right), 1, q, 10.0); ks.append(r.stats[kvmatch.STAT_KERNEL_NS] / 1e3)
        row.append(float(np.median(ks)))
    if len(row) < 3:
        print(f'L {L:2d} units {units} m {m:5d}: kernel us ' + '  '.join(f'f={f} {t:7.1f}' for f, t in zip(FS, row)), flush=True)
        continue
    a = (row[2] - row[0]) / 3.0
    ch = (m + 127) // 128
    print(f"L {L:2d} units {units} m {m:5d} chunks {ch:3d}: kernel us f=1 {row[0]:7.1f}  f=2 {row[1]:7.1f}  f=4 {row[2]:7.1f}   a = {a:5.2f} us/part   b = {(row[0] - a) / ch:5.2f} us/chunk", flush=True)
