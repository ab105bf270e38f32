#!/usr/bin/env python3
"""Class-C tiles or not (KVM_ED_HYBRID = 1: classes A / B only, 2: A / B / C) on lists below and around one unit per wave
slot, with the launcher's own split; RSM-ED, early abandon off; kernel time from HIP events (median of 8) and call time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
n = 100_000_000
s = datagen.gen_series(n)
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0); ctx.set_option(kvmatch.OPT_TIMING, 1)
eng = kvmatch.QueryEngine(ctx)
for m in (8192, 2048, 512):
    q = s[n // 3: n // 3 + m].copy()
    for c in (50_000, 100_000, 200_000, 350_000, 500_000, 1_000_000):
        for ml in (8.0, 16.0, 32.0, 64.0, 200.0):
            left, right = datagen.gen_intervals(n, m, c, ml, seed=int(ml) + c % 97)
            row = {}
            for rep in range(2):
                for mode in ("1", "2", "auto"):
                    if mode == "auto": os.environ.pop("KVM_ED_HYBRID", None)
                    else: os.environ["KVM_ED_HYBRID"] = mode
                    for _ in range(4): eng.verify((left, right), 1, q, 10.0)
                    ks, ws = [], []
                    for _ in range(8):
                        t0 = time.perf_counter(); r = eng.verify((left, right), 1, q, 10.0); ws.append((time.perf_counter() - t0) * 1e3)
                        ks.append(r.stats[kvmatch.STAT_KERNEL_NS] / 1e6)
                    row[mode] = (float(np.median(ks)), float(np.median(ws)))
            print(f"m {m:5d} cands {c:8d} mean run {ml:5.0f} ({len(left):6d} runs): A/B kernel {row['1'][0]:.4f} call {row['1'][1]:.4f}   A/B/C kernel {row['2'][0]:.4f} call {row['2'][1]:.4f}   library's choice kernel {row['auto'][0]:.4f}   A/B over A/B/C {row['1'][0] / row['2'][0]:.3f}   choice over the better {row['auto'][0] / min(row['1'][0], row['2'][0]):.3f}", flush=True)
