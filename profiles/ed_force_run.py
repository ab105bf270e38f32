#!/usr/bin/env python3
"""One RSM-ED (or cNSM-ED) query of the bench's shape with a forced launch shape of the ED pruner (KVM_OPT_ED_FORCE), every point
evaluated, a few repetitions: the program to put behind `rocprofv3 --pmc ... --` when a counter is wanted per form.
   ed_force_run.py <a16 0|1|2> <b16 0|1|2> [variant 0|1] [reps] [split]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen
a16, b16 = int(sys.argv[1]), int(sys.argv[2])
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
split = int(sys.argv[5]) if len(sys.argv) > 5 else 0
n, m = 100_000_000, 8192
series = datagen.gen_series(n)
qo = n // 3 + 5
q = series[qo - 1:qo - 1 + m].copy()
L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qo], seed=1)
with kvmatch.Phase2Context(0) as ctx:
    ctx.load_series(series)
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
    ctx.set_option(kvmatch.OPT_ED_FORCE, kvmatch.ed_force(a16=a16, b16=b16, split_a=split, split_b=split, dev_a16_any=1 if a16 == 2 else 0))
    ctx.set_option(kvmatch.OPT_TIMING, 1)
    eng = kvmatch.QueryEngine(ctx) if variant == 0 else kvmatch.NormQueryEngine(ctx)
    for _ in range(reps):
        a = eng.verify((L, R), 1, q, 10.0) if variant == 0 else eng.verify((L, R), 1, q, 10.0, 2.0, 1e9)
        print("kernel %.1f us  shape %s  answers %d" % (ctx._lib.kvm_last_kernel_ns(ctx._h) / 1e3, kvmatch.ed_shape(ctx.get_option(kvmatch.OPT_ED_LAST_SHAPE)), len(a)), flush=True)
