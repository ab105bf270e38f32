#!/usr/bin/env python3
"""Class A of the ED pruner in its two forms -- 8 candidates per lane (KVM_ED_A16=0) and 16 (KVM_ED_A16=1, round 5) -- under
every window-split pair (KVM_ED_SPLIT), against the launcher's own choice of form and pair ("auto"), over list shapes
(candidates x mean run length x m).  RSM-ED, early abandon off; kernel time from the dispatch's own stamps (median of 8).
One traced call per form prints the launcher's unit counts (stderr -> merge with 2>&1): input of fit_ed_a16.py."""
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
pairs = ("1,1", "1,2", "1,4", "2,1", "2,2", "2,4", "4,1", "4,2", "4,4")
shapes = [(m, c, ml) for m in (8192, 2048) for c in (200_000, 500_000, 1_000_000, 2_000_000, 4_000_000) for ml in (16.0, 64.0, 400.0)]
shapes += [(4096, c, ml) for c in (350_000, 1_500_000) for ml in (32.0, 150.0)] + [(1024, 1_000_000, 64.0), (1024, 4_000_000, 400.0)]
if len(sys.argv) > 1: shapes = shapes[::int(sys.argv[1])]
def t_of():
    for _ in range(3): eng.verify((left, right), 1, q, 10.0)
    return float(np.median([eng.verify((left, right), 1, q, 10.0).stats[kvmatch.STAT_KERNEL_NS] / 1e6 for _ in range(8)]))
for m, c, ml in shapes:
    q = s[n // 3: n // 3 + m].copy()
    left, right = datagen.gen_intervals(n, m, c, ml, seed=int(ml) + c % 97)
    for _ in range(6): eng.verify((left, right), 1, q, 10.0)   # (the clocks have dropped while the host generated the list)
    rows = {}
    for form in ("0", "1"):
        os.environ["KVM_ED_A16"] = form
        sys.stdout.flush(); os.environ.pop("KVM_ED_SPLIT", None); os.environ["KVM_TRACE"] = "1"
        eng.verify((left, right), 1, q, 10.0); os.environ.pop("KVM_TRACE")
        row = []
        for sp in pairs:
            os.environ["KVM_ED_SPLIT"] = sp
            row.append(t_of())
        rows[form] = row
    os.environ.pop("KVM_ED_SPLIT", None); os.environ.pop("KVM_ED_A16", None)
    auto = t_of()
    best = min(min(rows["0"]), min(rows["1"]))
    print(f"m {m:5d} cands {c:8d} mean run {ml:5.0f} ({len(left):6d} runs): " +
          " | ".join(f"form {8 if f == '0' else 16}: " + "  ".join(f"{sp} {t:.4f}" for sp, t in zip(pairs, rows[f])) for f in ("0", "1")) +
          f" | auto {auto:.4f}  best8 {min(rows['0']):.4f} best16 {min(rows['1']):.4f}  auto/best {auto / best:.3f}", flush=True)
