#!/usr/bin/env python3
"""CPU only: the oracle's index build (kvo_index_file: IndexBuilder.run + the file image) against the Python
transliteration (tests/java_transliteration.py: index_build_file) on hostile series -- NaN / Inf stretches, +-0, values on
the 0.05 key boundaries, 1e15 levels, denormals, plateaus, tiny n -- byte for byte.  `cpu_fuzz_index_restatements.py [cases] [seed]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle
import java_transliteration as J
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = total = 0
for case in range(cases):
    n = int(rng.choice([25, 26, 400, 3_000, 12_000]))
    kind = case % 8
    if kind == 0: s = np.cumsum(rng.normal(size=n)) * 0.05
    elif kind == 1: s = np.round(rng.normal(size=n) * 4) * 0.025
    elif kind == 2: s = rng.normal(size=n); s[rng.integers(0, n, max(1, n // 500))] = np.nan
    elif kind == 3: s = rng.normal(size=n); s[rng.integers(0, n, max(1, n // 800))] = np.inf; s[rng.integers(0, n, max(1, n // 800))] = -np.inf
    elif kind == 4: s = np.where(rng.random(n) < 0.5, -0.0, 0.0) + (rng.random(n) < 0.01) * rng.normal(size=n) * 0.01
    elif kind == 5: s = 1e15 + np.cumsum(rng.normal(size=n))
    elif kind == 6: s = rng.normal(size=n) * 5e-324 * rng.integers(0, 10, n)
    else: s = np.repeat(rng.normal(size=n // 300 + 1), 300)[:n]
    s = np.ascontiguousarray(s, dtype=np.float64)
    for w in (25, 50, 400):
        if n < w: continue
        total += 1
        want, winfo = oracle.index_file(s, w)
        got = J.index_build_file(s.tolist(), w)      # (NaN keys canonical on both sides, see its docstring)
        got = got[0] if isinstance(got, tuple) else got
        if bytes(got) != bytes(want):
            bad += 1
            print(f"DIFFER case {case} kind {kind} n={n} w={w}: {len(got)} vs {len(want)} bytes", flush=True)
print("index builds", total, "differences", bad)
sys.exit(1 if bad else 0)
