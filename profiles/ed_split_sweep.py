#!/usr/bin/env python3
"""The pruner's window-split pair (fA, fB) forced through KVM_ED_SPLIT against the launcher's own choice, over list shapes
(candidates x mean run length x m), RSM-ED, early abandon off; kernel time from HIP events (median of 8)."""
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
splits = ("1,1", "1,2", "1,4", "2,1", "2,2", "2,4", "4,1", "4,2", "4,4", "auto")  # auto last: the first
# configuration after a new list is built runs on a GPU whose clocks have dropped while the host generated the list
shapes = [(m, c, ml) for m in (8192, 1024) for c in (200_000, 500_000, 1_000_000, 2_000_000, 4_000_000) for ml in (16.0, 64.0, 400.0)]
if os.environ.get("SWEEP2"):  # a second, disjoint set (validation of the launcher's model)
    shapes = [(m, c, ml) for m in (4096, 2048, 512) for c in (100_000, 350_000, 700_000, 1_500_000, 3_000_000) for ml in (8.0, 32.0, 150.0)]
    shapes += [(8192, c, ml) for c in (350_000, 700_000, 1_500_000) for ml in (32.0, 150.0)]
if len(sys.argv) > 1: shapes = shapes[::int(sys.argv[1])]
for m, c, ml in shapes:
    q = s[n // 3: n // 3 + m].copy()
    left, right = datagen.gen_intervals(n, m, c, ml, seed=int(ml) + c % 97)
    if os.environ.get("SWEEP_UNITS"):   # one traced call: the launcher prints the unit counts of the three classes (stderr)
        sys.stdout.flush(); os.environ.pop("KVM_ED_SPLIT", None); os.environ["KVM_TRACE"] = "1"
        eng.verify((left, right), 1, q, 10.0); os.environ.pop("KVM_TRACE")
    row = []
    for sp in splits:
        if sp == "auto": os.environ.pop("KVM_ED_SPLIT", None)
        else: os.environ["KVM_ED_SPLIT"] = sp
        for _ in range(6 if sp == splits[0] else 3): eng.verify((left, right), 1, q, 10.0)
        ks = [eng.verify((left, right), 1, q, 10.0).stats[kvmatch.STAT_KERNEL_NS] / 1e6 for _ in range(8)]
        row.append(float(np.median(ks)))
    best = int(np.argmin(row[:-1]))
    print(f"m {m:5d} cands {c:8d} mean run {ml:5.0f} ({len(left):6d} runs): " + "  ".join(f"{sp} {t:.4f}" for sp, t in zip(splits, row)) + f"   best {splits[best]}  auto/best {row[-1] / row[best]:.3f}", flush=True)
