#!/usr/bin/env python3
"""Debug: N class-A tiles (runs of 40 candidates), a copy of the query planted in each; which tiles report, which twice?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 3_000_000, int(os.environ.get("DBG_M", "300"))
NT = int(os.environ.get("DBG_TILES", "64"))
s = datagen.gen_series(n, seed=99)
rng = np.random.default_rng(5)
q = s[1_000_000:1_000_000 + m].copy()
L = (1000 + 3000 * np.arange(NT)).astype(np.int32); R = (L + 39).astype(np.int32)
planted = []
for i in range(NT):
    off = int(L[i] + (i * 7) % 40)
    s[off - 1: off - 1 + m] = q + rng.normal(size=m) * 0.01; planted.append(off)
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
eng = kvmatch.QueryEngine(ctx)
for form in ("0", "1"):
    for sp in ("1,1", "2,1"):
        os.environ["KVM_ED_A16"] = form; os.environ["KVM_ED_SPLIT"] = sp
        got = eng.verify((L, R), 1, q, 2.0)
        ol = got.offsets.tolist()
        per = [ol.count(p) for p in planted]
        print(f"form {form} split {sp}: {len(ol)} answers; per tile:", "".join(str(c) for c in per), flush=True)
