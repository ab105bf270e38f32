#!/usr/bin/env python3
"""Multi-GPU readiness measured on ONE GPU (rounds 3-4; no 8-GPU node has ever run this code): the 8 shards that
kvmatch.distributed.split_intervals deals to the ranks of an 8-GPU job are verified one after the other on the one device,
each timed alone.  max / mean of the shard times = the load imbalance an 8-GPU run would see (its efficiency is at most
mean / max, before the all-gather), for BASELINE configs[1] (RSM-ED, every point evaluated), configs[3] (cNSM-DTW, whose DPs
cluster around the matches) and the configs[4]-style pair (RSM-DTW), weak (8e6 candidates, 1e6 per shard) and strong (1e6
candidates, 1.25e5 per shard).  Round 4 adds the STREAM view: every shard (and the unsharded list) also as a stream of 64
calls with 8 in flight through the native host threads (kvm_run_jobs) -- what a rank of the multi-GPU job sustains per query
when queries keep coming, which is what bench.py's sharded stream does; a lone call's latency (the round-3 figure) is the
other bound.  Writes gpurun_out/r04_shard_balance.json (kept under profiles/).   shard_balance.py [n] [world]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch
from kvmatch import datagen, distributed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
series = datagen.gen_series(n)
out = {"n": n, "world": world, "note": "one MI355X, shards timed one after the other (host API, one query in flight, median of 5); "
       "split_intervals default mode (whole intervals, interleaved)", "workloads": {}}
with kvmatch.Phase2Context(0) as ctx:
    ctx.load_series(series)
    qoff = n // 3 + 17
    workloads = [
        ("rsm_ed_m8192_every_point", 8192, 0, lambda c, iv, q: kvmatch.QueryEngine(c).verify(iv, 1, q, 10.0),
         lambda iv, q: kvmatch.Job(0, iv, 1, q, 10.0)),
        ("cnsm_dtw_m4096_rho204", 4096, 1, lambda c, iv, q: kvmatch.NormQueryEngineDtw(c).verify(iv, 1, q, 10.0, 0.05, 2.0, 10.0),
         lambda iv, q: kvmatch.Job(3, iv, 1, q, 10.0, rho=0.05, alpha=2.0, beta=10.0)),
        ("rsm_dtw_m4096_rho204", 4096, 1, lambda c, iv, q: kvmatch.QueryEngineDtw(c).verify(iv, 1, q, 10.0, 0.05),
         lambda iv, q: kvmatch.Job(2, iv, 1, q, 10.0, rho=0.05)),
    ]
    qs = kvmatch.QueryStream(ctx, in_flight=8)

    def stream_ms(job, calls=64):
        qs.run_jobs([job] * 8, collect=[])
        t = time.perf_counter()
        qs.run_jobs([job] * calls, collect=[])
        return (time.perf_counter() - t) / calls * 1e3

    for name, m, abandon, run, mkjob in workloads:
        q = series[qoff - 1: qoff - 1 + m].copy()
        qs.set_option(kvmatch.OPT_EARLY_ABANDON, abandon)
        res = {}
        for mode, total in (("weak", 1_000_000 * world), ("strong", 1_000_000)):
            L, R = datagen.gen_intervals(n, m, total, 64.0, must_cover=[qoff], seed=2)
            t_all = None
            shards = []
            for r in range(world):
                l, rr = distributed.split_intervals(L, R, world, r)
                run(ctx, (l, rr), q)
                ts = []
                for _ in range(5):
                    t = time.perf_counter()
                    a = run(ctx, (l, rr), q)
                    ts.append(time.perf_counter() - t)
                shards.append({"rank": r, "candidates": int((rr.astype(np.int64) - l + 1).sum()), "intervals": int(len(l)),
                               "ms": float(np.median(ts)) * 1e3, "stream_ms": stream_ms(mkjob((l, rr), q)), "dp": int(a.stats[kvmatch.STAT_PASS_LB_KEOGH2]), "answers": len(a)})
            ms = np.array([s["ms"] for s in shards])
            sms = np.array([s["stream_ms"] for s in shards])
            res[mode] = {"stream_max_ms": float(sms.max()), "stream_mean_ms": float(sms.mean()), "stream_max_over_mean": float(sms.max() / sms.mean()),
                         "shards": shards, "max_ms": float(ms.max()), "mean_ms": float(ms.mean()), "max_over_mean": float(ms.max() / ms.mean()),
                         "predicted_efficiency_8gpu": float(ms.mean() / ms.max())}
            if mode == "strong":   # the unsharded list on the one device, for the strong-scaling ceiling: t_1 / (8 max shard)
                run(ctx, (L, R), q)
                ts = []
                for _ in range(5):
                    t = time.perf_counter()
                    run(ctx, (L, R), q)
                    ts.append(time.perf_counter() - t)
                res[mode]["unsharded_ms"] = float(np.median(ts)) * 1e3
                res[mode]["predicted_speedup_8gpu"] = float(np.median(ts) * 1e3 / ms.max())
                res[mode]["unsharded_stream_ms"] = stream_ms(mkjob((L, R), q))
                res[mode]["predicted_stream_speedup_8gpu"] = float(res[mode]["unsharded_stream_ms"] / sms.max())
            print(name, mode, "one call: max/mean %.3f" % res[mode]["max_over_mean"], "shard ms", [round(x, 3) for x in ms.tolist()],
                  "| stream of 8 in flight: max/mean %.3f" % res[mode]["stream_max_over_mean"], "ms per query", [round(x, 3) for x in sms.tolist()],
                  ("| strong scaling x8 predicted: lone call %.2f, stream %.2f" % (res[mode]["predicted_speedup_8gpu"], res[mode]["predicted_stream_speedup_8gpu"])) if mode == "strong" else "", flush=True)
        out["workloads"][name] = res
    qs.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
    qs.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_shard_balance.json"), "w"), indent=1)   # (gpurun_out/ is what comes back
print("written gpurun_out/r04_shard_balance.json")                                                 #  from the GPU box: copied to profiles/)
