#!/usr/bin/env python3
"""What ONE rank of the multi-GPU job sustains per query through kvmatch.distributed.ShardedStream (device-resident
verification + pack kernel per slot, the calling thread doing the exchange and the host copy in query order), with the slots'
loops in native threads (kvm_run_device_jobs, round 4) against the Python slot threads of rounds 2-3: BASELINE configs[1]
(RSM-ED, m = 8192, every point evaluated), the unsharded 1e6-candidate list and shard 0 of 8 of it (strong scaling: 1.25e5
candidates), 8 in flight, 304 queries per run, world 1 (no collective: the exchange is the copy of the packed block).
Writes gpurun_out/r04_sharded_stream_rate.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen, distributed

n, m = 100_000_000, 8192
series = datagen.gen_series(n)
qoff = n // 3 + 17
q = series[qoff - 1: qoff - 1 + m].copy()
L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff], seed=2)
out = {"n": n, "m": m, "in_flight": 8, "queries_per_run": 304, "rows": []}
with kvmatch.Phase2Context(0) as ctx:
    ctx.load_series(series)
    for abandon in (0, 1):
        for label, (l, r) in (("unsharded_1e6", (L, R)), ("shard_0_of_8", distributed.split_intervals(L, R, 8, 0))):
            cands = int((r.astype(np.int64) - l + 1).sum())
            queries = [dict(valid_positions=(l, r), last_segment=1, query_data=q, epsilon=10.0)] * 304
            row = {"list": label, "candidates": cands, "early_abandon": abandon}
            ref = None
            for native in (False, True):
                ss = distributed.ShardedStream(ctx, 0, cands, in_flight=8, native=native)
                ss.set_option(kvmatch.OPT_EARLY_ABANDON, abandon)
                res = ss.run(queries[:16])
                ts = []
                for _ in range(3):
                    t = time.perf_counter()
                    res = ss.run(queries)
                    ts.append(time.perf_counter() - t)
                ss.close()
                if ref is None:
                    ref = res[-1]
                assert all(np.array_equal(a[0], ref[0]) and np.array_equal(a[1], ref[1]) for a in res)
                row["native_us_per_query" if native else "python_us_per_query"] = min(ts) / len(queries) * 1e6
            row["candidates_per_s_native"] = cands / (row["native_us_per_query"] * 1e-6)
            print(row, flush=True)
            out["rows"].append(row)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_sharded_stream_rate.json"), "w"), indent=1)
