#!/usr/bin/env python3
"""How much of the ED pruner's time is wave-slot quantisation?  Uniform interval lists (all runs `length` candidates,
`count` runs) on the bench series: 2048 wave units fill the chip exactly once (256 CUs x 8 waves)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 100_000_000, 8192
series = datagen.gen_series(n)
q = series[n // 3:n // 3 + m].copy()
ctx = kvmatch.Phase2Context(0); ctx.load_series(series); ctx.set_option(kvmatch.OPT_TIMING, 1)
ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
ctx.set_option(kvmatch.OPT_ED_KERNEL, int(os.environ.get('EDK', '0')))
eng = kvmatch.QueryEngine(ctx)
def run(length, count, label=""):
    stride = (n - m - length) // count
    L = (1 + np.arange(count, dtype=np.int64) * stride).astype(np.int32); R = (L + length - 1).astype(np.int32)
    eng.verify((L, R), 1, q, 10.0)
    ks = []
    for _ in range(5):
        a = eng.verify((L, R), 1, q, 10.0); ks.append(ctx._lib.kvm_last_kernel_ns(ctx._h) * 1e-6)
    k = min(ks)
    print(f"len {length:5d} x {count:6d} = {length*count:8d} cand: kernels {k:6.3f} ms  {length*count/k/1e6:8.1f} cand/us-ish  "
          f"{length*count*m*2/k/1e9:6.1f} TFLOP/s {label}")
for units in (1024, 2048, 2048 + 256, 3072, 4096, 4096 + 512, 6144, 8192, 16384):
    run(64, units * 8, f"[{units} A-units = {units/2048:.2f} rounds]")
for units in (2048, 3072, 4096):
    run(32, units * 8, f"[{units} B-units]")
run(512, 2048, "[8 A tiles per run]")
run(4096, 256, "")
run(48, 16384, "[A tiles 75% full]")
