#!/usr/bin/env python3
"""cNSM-ED (BASELINE configs[2]) query by query: the ten queries of bench.py's stream, each as a stream of its own (64 calls, 8
in flight, native host threads), early abandoning off and on, next to the number of candidates that pass the (alpha, beta)
filter -- what the stream's average is made of.  cnsm_ed_per_query.py [query indices, comma separated] [abandon modes] [in flight]
(one query, one mode, 1 in flight under rocprofv3 --kernel-trace --stats = that query's kernels)"""
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
QSEL = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else list(range(NQ))
MODES = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1]
F = int(sys.argv[3]) if len(sys.argv) > 3 else 8
with kvmatch.Phase2Context(0) as ctx:
    ctx.load_series(series)
    qs = kvmatch.QueryStream(ctx, in_flight=F)
    eng = kvmatch.NormQueryEngine(ctx)
    for abandon in MODES:
        qs.set_option(kvmatch.OPT_EARLY_ABANDON, abandon)
        tot = 0.0
        for j, q in [(j_, queries[j_]) for j_ in QSEL]:
            a = eng.verify((L, R), 1, q, 10.0, 2.0, 10.0)
            job = kvmatch.Job(1, (L, R), 1, q, 10.0, alpha=2.0, beta=10.0)
            qs.run_jobs([job] * 8, collect=[])
            t = time.perf_counter()
            qs.run_jobs([job] * 64, collect=[])
            us = (time.perf_counter() - t) / 64 * 1e6
            tot += us
            print("abandon %d query %d: pass filter %7d, answers %d, %.1f us per call in a stream of %d in flight" % (
                abandon, j, a.stats[kvmatch.STAT_PASS_FILTER] if hasattr(kvmatch, "STAT_PASS_FILTER") else -1, len(a), us, F), flush=True)
        print("abandon %d: mean %.1f us per call" % (abandon, tot / len(QSEL)), flush=True)
    qs.close()
