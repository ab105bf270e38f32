#!/usr/bin/env python3
"""CPU legs of bench.py on all host cores, run as a CHILD process (this process never touches the GPU): the oracle's loops
(C ports of the reference's phase-2 blocks) over inputs bench.py hands over in an .npz file, interval shards spread over
forked worker processes.  TEST / MEASUREMENT INFRASTRUCTURE ONLY -- lives under oracle/.

  cpu_baseline.py rsm_ed <n> <eps> <workers> <in.npz>
      times K/QueryEngine.java:338-363 (early abandon off, then on) over `left` / `right` of in.npz; prints one JSON line
  cpu_baseline.py cnsm_ed_stream <n> <eps> <workers> <in.npz> <alpha> <beta>
      times K/NormQueryEngine.java:432-528 over the queries of a stream: in.npz holds qs (NQ x m), the NL interval lists
      concatenated (left / right / list_off) and q_list (the list of each query); one core over the first `one_core_queries`
      of them, then all of them on `workers` cores (whole intervals dealt to the workers: the running sums restart per
      interval), early abandon off and on; prints one JSON line
  cpu_baseline.py cnsm_dtw_answers <n> <eps> <workers> <in.npz> <out.npz> <rho> <alpha> <beta>
      the answers of K/NormQueryEngineDtw.java:457-603 for the checker (`parity_vs_oracle` of the cNSM-DTW workload): whole
      intervals dealt round-robin to the workers (running sums restart per interval, so no value depends on the deal; the
      fetched chunks -- hence the clipping of the data envelope of LB_Keogh -- do, which can only matter at an exact tie
      of a lower bound with eps^2), merged and sorted like the reference sorts them; writes off / dist / seconds to out.npz
"""
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


def _work_ed(args):
    import oracle
    lo, hi, no_abandon = args
    off, dist, st = oracle.rsm_ed(_G["series"], _G["q"], _G["L"][lo:hi], _G["R"][lo:hi], 0, _G["eps"], no_abandon=no_abandon)
    return st["windows"], len(off)


def _work_cnsm_ed(args):
    import oracle
    j, lo, hi, no_abandon = args
    a, b = _G["list_off"][_G["q_list"][j]], _G["list_off"][_G["q_list"][j] + 1]
    L, R = _G["L"][a:b][lo:hi], _G["R"][a:b][lo:hi]
    if len(L) == 0:
        return 0, 0, 0
    off, dist, st = oracle.cnsm_ed(_G["series"], _G["qs"][j], L, R, 0, _G["eps"], _G["alpha"], _G["beta"], no_abandon=no_abandon)
    return int((R.astype(np.int64) - L + 1).sum()), len(off), st["pass_filter"]


def _work_dtw(w):
    import oracle
    L, R = _G["L"][w::_G["workers"]], _G["R"][w::_G["workers"]]
    if len(L) == 0:
        return np.empty(0, np.int32), np.empty(0, np.float64)
    off, dist, st = oracle.cnsm_dtw(_G["series"], _G["q"], L, R, 0, _G["eps"], _G["rho"], _G["alpha"], _G["beta"])
    return off, dist


def main():
    mode, n, eps, workers, fin = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    from kvmatch import datagen  # host-side generators only (no HIP call is made in this process)
    z = np.load(fin)
    series = datagen.gen_series(n)
    q = z["q"] if "q" in z else z["qs"][0]
    L, R = z["left"], z["right"]
    m = len(q)
    # bench.py plants copies of its queries into the series it stages; the slices this leg reads must be the staged ones:
    # the planted windows are handed over with the inputs
    if "patch_pos" in z:
        for p, w in zip(z["patch_pos"].tolist(), z["patch_val"]):
            series[p - 1: p - 1 + len(w)] = w
    _G.update(series=series, q=q, L=L, R=R, eps=eps, workers=workers)
    if mode == "rsm_ed":
        k = len(L)
        bounds = [(k * i // workers, k * (i + 1) // workers) for i in range(workers)]
        out = {"cores": workers}
        with mp.get_context("fork").Pool(workers) as pool:
            pool.map(_work_ed, [(0, 1, 1)] * workers)  # warm the workers (library load)
            for name, flag in (("no_abandon", 1), ("early_abandon", 0)):
                t = time.perf_counter()
                res = pool.map(_work_ed, [(lo, hi, flag) for lo, hi in bounds])
                dt = time.perf_counter() - t
                out[name] = {"candidates_per_s": sum(r[0] for r in res) / dt, "seconds": dt, "answers": sum(r[1] for r in res)}
        print(json.dumps(out))
    elif mode == "cnsm_ed_stream":
        alpha, beta = float(sys.argv[6]), float(sys.argv[7])
        qs, list_off, q_list = z["qs"], z["list_off"], z["q_list"]
        _G.update(qs=qs, list_off=list_off, q_list=q_list, alpha=alpha, beta=beta)
        nq = len(qs)
        n_one = int(z["one_core_queries"]) if "one_core_queries" in z else nq
        klen = [int(list_off[q_list[j] + 1] - list_off[q_list[j]]) for j in range(nq)]
        out = {"cores": workers, "queries": nq, "one_core_queries": n_one}
        for name, flag in (("no_abandon", 1), ("early_abandon", 0)):   # one core, in this process
            t = time.perf_counter()
            res = [_work_cnsm_ed((j, 0, klen[j], flag)) for j in range(n_one)]
            dt = time.perf_counter() - t
            out["one_core_" + name] = {"candidates_per_s": sum(r[0] for r in res) / dt, "seconds": dt, "answers": sum(r[1] for r in res),
                                       "pass_filter": sum(r[2] for r in res)}
        tasks = lambda flag: [(j, klen[j] * i // workers, klen[j] * (i + 1) // workers, flag) for j in range(nq) for i in range(workers)]  # noqa: E731
        with mp.get_context("fork").Pool(workers) as pool:
            pool.map(_work_cnsm_ed, [(0, 0, 1, 0)] * workers)  # warm the workers (library load)
            for name, flag in (("no_abandon", 1), ("early_abandon", 0)):
                t = time.perf_counter()
                res = pool.map(_work_cnsm_ed, tasks(flag), chunksize=1)
                dt = time.perf_counter() - t
                out[name] = {"candidates_per_s": sum(r[0] for r in res) / dt, "seconds": dt, "answers": sum(r[1] for r in res),
                             "pass_filter": sum(r[2] for r in res)}
        print(json.dumps(out))
    elif mode == "cnsm_dtw_answers":
        fout, rho, alpha, beta = sys.argv[6], int(sys.argv[7]), float(sys.argv[8]), float(sys.argv[9])
        _G.update(rho=rho, alpha=alpha, beta=beta)
        t = time.perf_counter()
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(_work_dtw, list(range(workers)))
        dt = time.perf_counter() - t
        off = np.concatenate([r[0] for r in res]).astype(np.int32)
        dist = np.concatenate([r[1] for r in res]).astype(np.float64)
        order = np.lexsort((off, dist))  # stable sort by distance of an offset-ascending list (K/QueryEngine.java:373)
        np.savez(fout, off=off[order], dist=dist[order], seconds=dt)
        print(json.dumps({"cores": workers, "seconds": dt, "answers": int(len(off))}))
    else:
        raise SystemExit(f"unknown mode {mode}")


if __name__ == "__main__":
    main()
