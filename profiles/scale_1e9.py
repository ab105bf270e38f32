"""BASELINE configs[4] scale check: n = 1e9 points (8 GB series + 16 GB prefix sums in HBM), all four variants on
phase-1-shaped candidate lists spread over the whole series, compared bit for bit with the oracle."""
import json, os, sys, time
import numpy as np, psutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kvmatch
from kvmatch import datagen
import oracle

n = int(os.environ.get("N", 1_000_000_000))
avail = psutil.virtual_memory().available
print("host memory available: %.1f GB" % (avail / 2**30), flush=True)
if avail < n * 8 * 2.5:
    print("not enough host memory for the check"); sys.exit(0)
out = {"n": n}
t0 = time.perf_counter(); s = datagen.gen_series(n); out["gen_s"] = time.perf_counter() - t0
ctx = kvmatch.Phase2Context(0)
qoff = n - 123_456_789 if n > 2 * 123_456_789 else n // 2

def same(got, want):
    return bool(len(got) == len(want[0]) and np.array_equal(got.offsets, want[0]) and np.array_equal(got.distances, want[1]))

def timed(fn, reps=3):
    fn(); best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t)
    return r, best

for m, cand in ((8192, 1_000_000), (4096, 100_000)):
    planted = [1, n // 2, n - m + 1 - 10]
    q = datagen.plant_copies(s, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.02, -0.03], 0.01)
    t0 = time.perf_counter(); ctx.load_series(s); out[f"load_s_m{m}"] = time.perf_counter() - t0
    L, R = datagen.gen_intervals(n, m, cand, 64.0, must_cover=[qoff] + planted, seed=1)
    out[f"intervals_m{m}"] = int(len(L)); out[f"last_offset_m{m}"] = int(R[-1])
    if m == 8192:
        r, dt = timed(lambda: kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 10.0))
        out["rsm_ed"] = {"ms": dt * 1e3, "cand_per_s": r.cnt_candidate / dt, "answers": len(r), "parity": same(r, oracle.rsm_ed(s, q, L, R, 0, 10.0))}
        r, dt = timed(lambda: kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 3.0, 2.0, 10.0))
        out["cnsm_ed"] = {"ms": dt * 1e3, "cand_per_s": r.cnt_candidate / dt, "answers": len(r), "parity": same(r, oracle.cnsm_ed(s, q, L, R, 0, 3.0, 2.0, 10.0))}
    else:
        r, dt = timed(lambda: kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, 10.0, 0.05))
        out["rsm_dtw"] = {"ms": dt * 1e3, "cand_per_s": r.cnt_candidate / dt, "answers": len(r), "parity": same(r, oracle.rsm_dtw(s, q, L, R, 0, 10.0, 204))}
        r, dt = timed(lambda: kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 3.0, 0.05, 2.0, 10.0))
        out["cnsm_dtw"] = {"ms": dt * 1e3, "cand_per_s": r.cnt_candidate / dt, "answers": len(r), "parity": same(r, oracle.cnsm_dtw(s, q, L, R, 0, 3.0, 204, 2.0, 10.0))}
    print(json.dumps(out), flush=True)
# index rows + TimeSeriesOperator at the far end of the series
got = ctx.read_time_series(n - 999, 1000)
out["read_tail_ok"] = bool(np.array_equal(got, s[n - 1000:]))
t0 = time.perf_counter(); keys, first, last = ctx.index_rows(50); out["index_rows_w50_s"] = time.perf_counter() - t0
out["index_rows_w50_intervals"] = int(len(keys)); out["index_rows_cover"] = bool(first[0] == 1 and last[-1] == n - 49)
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "scale_1e9.json"), "w"), indent=1)
