#!/usr/bin/env python3
"""Where does a stream of sharded queries spend its time at world 1?  The bench's RSM-ED workload (n = 1e8, m = 8192, 1e6
candidates per query) through kvm_run_sharded_jobs on a group of this one GPU, with and without the RCCL exchange, early
abandoning on and off, several depths -- next to the host path (kvm_run_jobs); prints kvm_group_stat per configuration.
  python profiles/tools/group_stream_probe.py [--n N] [--queries Q] [--only host|group|group_rccl] [--depths 8,16] [--abandon 0,1]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np  # noqa: E402
import kvmatch  # noqa: E402
from kvmatch import datagen  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100_000_000)
ap.add_argument("--queries", type=int, default=600)
ap.add_argument("--only", default="")
ap.add_argument("--depths", default="8,16")
ap.add_argument("--abandon", default="0,1")
ap.add_argument("--variant", default="rsm_ed", choices=["rsm_ed", "cnsm_ed"], help="cnsm_ed: alpha = 2, beta = 10 (BASELINE configs[2])")
a = ap.parse_args()
n, nq, m = a.n, a.queries, 8192
depths = [int(x) for x in a.depths.split(",")]
abandon = [int(x) for x in a.abandon.split(",")]
series = datagen.gen_series(n)
qo = [n // 7 + 1, n // 3 + 5, n // 2 + 11]
queries = [series[o - 1:o - 1 + m].copy() for o in qo]
lists = [datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=qo, seed=1 + i) for i in range(4)]
cands = [int((r.astype(np.int64) - l + 1).sum()) for l, r in lists]
if a.variant == "cnsm_ed":
    jobs = [kvmatch.Job(1, lists[t % 4], 1, queries[t % 3], 10.0, alpha=2.0, beta=10.0) for t in range(nq)]
else:
    jobs = [kvmatch.Job(0, lists[t % 4], 1, queries[t % 3], 10.0) for t in range(nq)]
total = sum(cands[t % 4] for t in range(nq))
if a.only in ("", "host"):
    with kvmatch.Phase2Context(0) as ctx:
        ctx.load_series(series)
        for F in depths[:1]:
            with kvmatch.QueryStream(ctx, F) as qs:
                for ab in abandon:
                    qs.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
                    qs.run_jobs(jobs[:64], collect=[])
                    t0 = time.perf_counter()
                    qs.run_jobs(jobs, collect=[])
                    dt = time.perf_counter() - t0
                    print(json.dumps({"path": "host (kvm_run_jobs)", "in_flight": F, "early_abandon": ab, "cands_per_s": total / dt,
                                      "us_per_query": dt / nq * 1e6}), flush=True)
for rccl in (False, True):
    if a.only not in ("", "group_rccl" if rccl else "group"):
        continue
    for F in depths:
        with kvmatch.DeviceGroup(devices=[0], in_flight=F, rccl_always=rccl) as g:
            g.load_series(series)
            for ab in abandon:
                g.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
                g.run_jobs(jobs[:64], collect=[])
                t0 = time.perf_counter()
                g.run_jobs(jobs, collect=[])
                dt = time.perf_counter() - t0
                print(json.dumps({"path": "group", "rccl": rccl, "in_flight": F, "early_abandon": ab, "cands_per_s": total / dt,
                                  "us_per_query": dt / nq * 1e6, "calling_thread": g.stats()}), flush=True)
