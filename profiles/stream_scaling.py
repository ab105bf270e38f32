"""Rate of a stream of latency-bound calls against the number of queries in flight: RSM-ED with the reference's early
abandoning and cNSM-ED on the bench list (n = 1e8, m = 8192, 1e6 candidates), 10 queries x 4 lists as in bench.py.
stream_scaling.py [in_flight ...]   (STREAM_CALLS: calls per measurement, default 400; STREAM_ONLY=rsm|cnsm)"""
import os, sys, time
sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n, m = 100_000_000, 8192
flights = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 5, 8]
calls = int(os.environ.get("STREAM_CALLS", "400")); only = os.environ.get("STREAM_ONLY", "")
s = datagen.gen_series(n)
rng = np.random.default_rng(5)
qoffs = [int(x) for x in rng.integers(1, (n - m) // (4 * m), 10) * 4 * m + 1]
queries = [s[o - 1: o - 1 + m].copy() for o in qoffs]
lists = [datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=qoffs, seed=11 + j) for j in range(4)]
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
for name, mk in (("rsm_ed abandon", lambda i: (lambda c: kvmatch.QueryEngine(c).verify(lists[i % 4], 1, queries[i % 10], 10.0))),
                 ("cnsm_ed abandon", lambda i: (lambda c: kvmatch.NormQueryEngine(c).verify(lists[i % 4], 1, queries[i % 10], 10.0, 2.0, 10.0)))):
    if only and only not in name: continue
    for F in flights:
        qs = kvmatch.QueryStream(ctx, in_flight=F)
        qs.run([mk(i) for i in range(60)])
        t0 = time.perf_counter(); r = qs.run([mk(i) for i in range(calls)]); dt = time.perf_counter() - t0
        cand = sum(x.cnt_candidate for x in r)
        print(f"{name}: in flight {F}: {dt / calls * 1e6:7.1f} us per call, {cand / dt:.3e} candidates/s, answers {sum(len(x) for x in r)}", flush=True)
        qs.close()
