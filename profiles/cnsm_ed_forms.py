#!/usr/bin/env python3
"""cNSM-ED (BASELINE configs[2]): the ED pruner's launch shape on DEVICE-BUILT lists, forced through KVM_OPT_ED_FORCE -- the
in-kernel rule (default) against never split / 16 per lane at any size / both; per query: kernel time of one call alone
(HIP events) and microseconds per call in a stream of 8 in flight.   cnsm_ed_forms.py [queries] [abandon]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen

n, m, NQ = 100_000_000, 8192, 10
series = datagen.gen_series(n)
rng = np.random.default_rng(0x51EED + m)
cells = rng.choice(max(3 * NQ + 2, n // (4 * m) - 2), size=3 * NQ, replace=False) + 1
pos = [int(c) * 4 * m + 1 for c in cells]
queries, must = [], []
for j in range(NQ):
    qo, planted = pos[3 * j], [pos[3 * j + 1], pos[3 * j + 2]]
    queries.append(datagen.plant_copies(series, qo, m, planted, [1.0, 1.0], [0.02, -0.03], 0.01, seed=100 + j))
    must += [qo] + planted
L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=sorted(set(must)), seed=1)
QSEL = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 6, 3, 1, 9]
AB = int(sys.argv[2]) if len(sys.argv) > 2 else 0
F = kvmatch.ed_force
FORMS = [("default", 0), ("no split", F(dev_no_split=1)), ("16/lane any size", F(dev_a16_any=1)), ("16/lane, no split", F(dev_a16_any=1, dev_no_split=1)),
         ("8/lane only", F(a16=1)), ("8/lane, no split", F(a16=1, dev_no_split=1))]
with kvmatch.Phase2Context(0) as ctx:
    ctx.load_series(series)
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, AB)
    eng = kvmatch.NormQueryEngine(ctx)
    for j in QSEL:
        q = queries[j]
        for name, force in FORMS:
            ctx.set_option(kvmatch.OPT_ED_FORCE, force)
            ctx.set_option(kvmatch.OPT_TIMING, 1)
            ns = []
            for rep in range(6):
                a = eng.verify((L, R), 1, q, 10.0, 2.0, 10.0)
                if rep >= 2: ns.append(ctx._lib.kvm_last_kernel_ns(ctx._h))
            ctx.set_option(kvmatch.OPT_TIMING, 0)
            qs = kvmatch.QueryStream(ctx, in_flight=8)
            job = kvmatch.Job(1, (L, R), 1, q, 10.0, alpha=2.0, beta=10.0)
            qs.run_jobs([job] * 16, collect=[])
            t = time.perf_counter()
            qs.run_jobs([job] * 96, collect=[])
            us = (time.perf_counter() - t) / 96 * 1e6
            qs.close()
            print("query %d pass %7d  %-20s kernel alone %7.1f us   stream of 8: %6.1f us per call" % (j, a.stats[kvmatch.STAT_PASS_FILTER], name, np.mean(ns) / 1e3, us), flush=True)
