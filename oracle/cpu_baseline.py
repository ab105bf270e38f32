#!/usr/bin/env python3
"""CPU baseline on all host cores: the oracle's RSM-ED loop (a C port of K/QueryEngine.java:338-363) over the bench
workload, interval shards spread over `workers` forked processes.  Run as a CHILD process by bench.py (this process
never touches the GPU).  TEST/MEASUREMENT INFRASTRUCTURE ONLY -- lives under oracle/.  Prints one JSON line."""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

_G = {}


def _work(args):
    import oracle
    lo, hi, no_abandon = args
    off, dist, st = oracle.rsm_ed(_G["series"], _G["q"], _G["L"][lo:hi], _G["R"][lo:hi], 0, _G["eps"], no_abandon=no_abandon)
    return st["windows"], len(off)


def main():
    n, m, cands, eps, workers = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
    from kvmatch import datagen  # host-side generators only (no HIP call is made in this process)
    series = datagen.gen_series(n)
    qoff = n // 3 + 17
    planted = [n // 7, n // 2 + 1000, n - 3 * m]
    q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.02, -0.03], 0.01)
    L, R = datagen.gen_intervals(n, m, cands, 64.0, must_cover=[qoff] + planted, seed=1)
    _G.update(series=series, q=q, L=L, R=R, eps=eps)
    k = len(L)
    bounds = [(k * i // workers, k * (i + 1) // workers) for i in range(workers)]
    out = {"cores": workers}
    with mp.get_context("fork").Pool(workers) as pool:
        pool.map(_work, [(0, 1, 1)] * workers)  # warm the workers (library load)
        for name, flag in (("no_abandon", 1), ("early_abandon", 0)):
            t = time.perf_counter()
            res = pool.map(_work, [(lo, hi, flag) for lo, hi in bounds])
            dt = time.perf_counter() - t
            out[name] = {"candidates_per_s": sum(r[0] for r in res) / dt, "seconds": dt, "answers": sum(r[1] for r in res)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
