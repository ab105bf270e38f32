"""Pure class-B and class-C lists for the PMC passes of profiles/pmc_classes.sh: 31 250 runs of 32 / 62 500 runs of 16 candidates."""
import sys; sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 100_000_000, 8192
s = datagen.gen_series(n); q = s[n // 3:n // 3 + m].copy()
ctx = kvmatch.Phase2Context(0); ctx.load_series(s); ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
which = sys.argv[1]
L, tiles = {"A": (64, 15625), "B": (32, 31250), "C": (16, 62500)}[which]
st = 1 + np.arange(tiles, dtype=np.int64) * ((n - m - 100) // tiles)
for _ in range(300):
    kvmatch.QueryEngine(ctx).verify((st.astype(np.int32), (st + L - 1).astype(np.int32)), 1, q, 10.0)
