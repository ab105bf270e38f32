"""Wall-clock checks (-m gpu).  The file name sorts behind every parity file on purpose: `pytest -x` reaches these only after
the parity suite has run, so a slow box of the pool cannot hide parity results behind a timing assertion."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 100_000_000


@pytest.fixture(scope="module")
def big(built):
    import kvmatch
    from kvmatch import datagen
    series = datagen.gen_series(N)
    m = 8192
    qoff = N // 3 + 17
    planted = [N // 7, N // 2 + 1000, N - 3 * m]
    q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.02, -0.03], 0.01)
    ctx = kvmatch.Phase2Context(0)
    ctx.load_series(series)
    yield kvmatch, datagen, series, ctx, q, qoff, planted
    ctx.close()


def _fit_files():
    import glob
    import json
    import os
    prof = os.path.join(os.path.dirname(__file__), "..", "profiles")
    files = sorted(glob.glob(os.path.join(prof, "r[0-9][0-9]_phase2_cost_model.json")))
    assert files, "profiles/rNN_phase2_cost_model.json (written by profiles/fit_cost_model.py) is missing"
    fit = json.load(open(files[-1]))
    held = files[-1].replace("_cost_model.json", "_held_out.json")
    return fit, (json.load(open(held)) if os.path.exists(held) else None), held


BOX_SPREAD = 0.08   # the boxes of the pool differ by about this much on the same call


def _fit_limits(fit, held):
    """variant -> 1.3 x the model's worst residual -- over the points of its own fit AND over the held-out shapes below as measured
    when the constants were written (profiles/rNN_phase2_held_out.json) -- plus the spread between boxes."""
    lim = {}
    for v, k in enumerate(("rsm_ed", "cnsm_ed", "rsm_dtw", "cnsm_dtw")):
        worst = float(fit[k]["max_rel_err_of_prediction"])
        for h in (held or {}).get("shapes", []):
            if h["variant"] == v:
                worst = max(worst, abs(h["estimate_ms"] - h["measured_ms"]) / h["measured_ms"])
        lim[v] = 1.3 * worst + BOX_SPREAD
    return lim


def test_phase2_time_model_against_measured_calls(big):
    """kvm_phase2_time_estimate_ms (constants compiled in from csrc/kvm_cost_model.inc, written by profiles/fit_cost_model.py)
    against calls measured here, on shapes of the regime the fit covers (n = 1e8, selective queries, early abandon on) but
    NOT its grid points: other interval seeds, other run lengths, m = 3072 between its lengths.  The fit's own residuals are
    <= 35 % for the ED engines and <= 42 % for the DTW engines (profiles/r06_phase2_cost_model.json: refitted in round 6).  The
    table is printed; what fails the run is at its end.  The DTW engines' time depends on how many candidates reach the DP -- 3 or 700 for the same
    number of offsets, decided by the data under the intervals, which no a-priori model knows (this test's shape of 8e5
    offsets at m = 4096 takes 0.96 ms where the grid's 1e6 offsets took 1.2 - 2.5 ms; the model says 1.24).  A change of
    the launch chain or of a kernel that moves calls by more than that fails here, so the constants cannot silently rot."""
    import time
    kvmatch, datagen, series, ctx, q8, qoff, planted = big
    shapes = [(0, 3072, 200_000, 16.0), (0, 8192, 2_000_000, 64.0), (0, 2048, 3_000_000, 256.0), (1, 4096, 500_000, 32.0),
              (1, 8192, 2_500_000, 128.0), (2, 2048, 300_000, 32.0), (2, 4096, 800_000, 128.0), (3, 3072, 300_000, 64.0),
              (3, 4096, 600_000, 256.0)]
    def measure(shape, reps, pick):
        variant, m, offsets, mean_len = shape
        q = q8[:m].copy()                         # (the head of the planted query: its copies match on the same heads)
        L, R = datagen.gen_intervals(N, m, offsets, mean_len, must_cover=[qoff] + planted, seed=31 + m)
        e = 10.0 * (m / 8192.0) ** 0.5
        eng, args = [(kvmatch.QueryEngine(ctx), (e,)), (kvmatch.NormQueryEngine(ctx), (e, 1.5, 5.0)),
                     (kvmatch.QueryEngineDtw(ctx), (e, 0.05)), (kvmatch.NormQueryEngineDtw(ctx), (e, 0.05, 1.5, 5.0))][variant]
        ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)   # timing: the cost does not depend on whether JDK's sort would throw
        try:
            eng.verify((L, R), 1, q, *args)
            ts = []
            for _ in range(reps):
                t = time.perf_counter(); a = eng.verify((L, R), 1, q, *args); ts.append(time.perf_counter() - t)
        finally:
            ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 0)
        got = float(pick(ts)) * 1e3
        est = kvmatch.phase2_time_estimate_ms(variant, len(L), offsets, m)
        row = (abs(est - got) / got, variant, m, offsets, mean_len, round(got, 3), round(est, 3), len(a))
        dp_row = None
        if variant >= 2:   # the DTW engines, with the DP count the finished call reports: the form that knows what the data decided
            est_dp = kvmatch.phase2_time_estimate_dp_ms(variant, len(L), offsets, m, a.stats[kvmatch.STAT_PASS_LB_KEOGH2])
            dp_row = (abs(est_dp - got) / got, variant, m, offsets, a.stats[kvmatch.STAT_PASS_LB_KEOGH2], round(got, 3), round(est_dp, 3))
        return row, dp_row

    worst, with_dp = [], []
    for shape in shapes:
        row, dp_row = measure(shape, 7, np.median)
        worst.append(row)
        if dp_row:
            with_dp.append(dp_row)
    print("\nphase-2 time model against measured calls (relative error, variant, m, offsets, mean run, measured ms, estimate ms, answers):")
    for w in sorted(worst, reverse=True):
        print("   %.3f  v%d m=%d offsets=%d run=%.0f  %.3f ms vs %.3f ms  (%d answers)" % w)
    # Two guards.  (1) The model: the run fails beyond 1.3 x the model's worst residual (its own fit and these shapes as measured
    # when the constants were written) + the spread between boxes -- r06: 54 / 30 / 51 / 62 %.  The DTW engines' time follows the
    # number of DPs the data decides, so their a-priori limit stays wide; the form that is GIVEN the DP count is held to 45 %.
    # (2) The calls themselves: every shape against its time in profiles/rNN_phase2_held_out.json, measured when the constants
    # were written -- more than 1.3 x (+ the box spread) slower fails, whatever the model says: a regression of the launch chain or
    # of a kernel cannot hide behind the model's tolerance (rounds 4 - 5: twice the worst residual, which let a 1.8 x slower DTW
    # call pass).  KVM_WRITE_HELD_OUT=1 rewrites the file from this run (after a refit).
    import json
    import os
    fit, held, held_path = _fit_files()
    if os.environ.get("KVM_WRITE_HELD_OUT") == "1":
        held = {"shapes": [{"variant": w[1], "m": w[2], "offsets": w[3], "mean_run": w[4], "measured_ms": w[5], "estimate_ms": w[6]} for w in worst],
                "note": "tests/test_zz_perf_gpu.py::test_phase2_time_model_against_measured_calls with KVM_WRITE_HELD_OUT=1"}
        json.dump(held, open(held_path, "w"), indent=1)
    limits = _fit_limits(fit, held)
    print("limits (1.3 x the model's worst residual + box spread, per variant):", {k: round(v, 3) for k, v in limits.items()})
    ref = {(h["variant"], h["m"], h["offsets"]): h["measured_ms"] for h in (held or {}).get("shapes", [])}

    def out_of_bounds(w):
        r = ref.get((w[1], w[2], w[3]))
        return w[0] > limits[w[1]] or (r is not None and w[5] > (1.3 + BOX_SPREAD) * r)

    # A box of the pool is shared: a burst of somebody else's work has made a 0.24-ms call read 0.38 ms in the median of 7.  A
    # shape that is out of bounds at first sight is measured once more, 21 calls, lower quartile -- noise only ever adds time,
    # a regression of the launch chain or of a kernel is still there the second time.
    for i, w in enumerate(worst):
        if out_of_bounds(w):
            row, dp_row = measure(shapes[i], 21, lambda ts: np.percentile(ts, 25))
            print("   measured again: v%d m=%d offsets=%d  %.3f ms at first, %.3f ms the second time" % (w[1], w[2], w[3], w[5], row[5]))
            worst[i] = row
            if dp_row:
                with_dp[[k for k, d in enumerate(with_dp) if d[1:4] == dp_row[1:4]][0]] = dp_row
    bad = [w for w in worst if w[0] > limits[w[1]]]
    assert not bad, (bad, worst)
    print("the DTW engines with the DP count as an argument (relative error, variant, m, offsets, DPs, measured ms, estimate ms):")
    for w in sorted(with_dp, reverse=True):
        print("   %.3f  v%d m=%d offsets=%d dp=%d  %.3f ms vs %.3f ms" % w)
    assert not [w for w in with_dp if w[0] > 0.45], with_dp
    assert held is not None, "profiles/rNN_phase2_held_out.json is missing (run once with KVM_WRITE_HELD_OUT=1 after a refit)"
    slow = [(w[5] / ref[(w[1], w[2], w[3])], w) for w in worst if w[5] > (1.3 + BOX_SPREAD) * ref[(w[1], w[2], w[3])]]
    print("measured / recorded time per shape:", sorted(round(w[5] / ref[(w[1], w[2], w[3])], 3) for w in worst))
    assert not slow, slow
