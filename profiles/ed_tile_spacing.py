#!/usr/bin/env python3
"""Kernel time of the ED pruner on 15 625 isolated full class-A tiles (runs of 64 candidates) as a function of the SPACING of
the runs in the series: from 80 points (the rows of a unit read the same cache lines a little apart in time) to 6 400 (the
whole series) -- what about memory locality costs the isolated tiles of profiles/ed_run_length.py their 33 %?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
n, m, L, runs = 100_000_000, 8192, 64, 15625
s = datagen.gen_series(n)
q = s[n // 3: n // 3 + m].copy()
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0); ctx.set_option(kvmatch.OPT_TIMING, 1)
eng = kvmatch.QueryEngine(ctx)
WARM = 3000
for spacing in [int(x) for x in os.environ.get("SPACINGS", "66,80,128,256,512,1024,2048,4096,6390").split(",")]:
    starts = 1 + np.arange(runs, dtype=np.int64) * spacing
    left = starts.astype(np.int32); right = (starts + L - 1).astype(np.int32)
    for _ in range(WARM): r = eng.verify((left, right), 1, q, 10.0)   # (the first configuration also brings the clocks up: ~1 s of load)
    WARM = 5
    ks = []
    for _ in range(20):
        r = eng.verify((left, right), 1, q, 10.0); ks.append(r.stats[kvmatch.STAT_KERNEL_NS] / 1e6)
    k = float(np.median(ks))
    print(f"runs of {L} every {spacing:5d} points ({spacing * 8:6d} B; footprint {(runs * spacing + m) * 8 / 1e6:7.1f} MB): kernel {k:.4f} ms  {2 * m * runs * L / k / 1e9:.1f} TFLOP/s", flush=True)
