"""Adversarial inputs (-m gpu): see tests/adversarial.py.  Every verification must equal the oracle bit for bit --
including the reference's lower-bound ties (DtwUtils.dtw, K/utils/DtwUtils.java:324-326), which integer-valued data with
an integer eps^2 does hit."""
import pytest

import adversarial
import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.strict_sort
@pytest.mark.parametrize("dp_form", [0, 3])   # 3: every DTW list through the packed register form of the DP
def test_adversarial_inputs_all_variants(built, dp_form):
    import kvmatch
    from kvmatch import datagen
    ctx = kvmatch.Phase2Context(0)
    ctx.set_option(kvmatch.OPT_DP_KERNEL, dp_form)
    lines = []
    total, bad, ties = adversarial.run(kvmatch, datagen, oracle, ctx, rounds=1, seed=11, log=lines.append)
    ctx.close()
    assert bad == 0, "\n".join(l for l in lines if l.startswith("MISMATCH"))
    assert total >= 700
    assert ties > 0           # integer-valued data with an integer eps^2 does reach the sequential tie path


@pytest.mark.parametrize("dp_form", [0, 2, 3])
def test_a_row_bound_exactly_at_eps2_is_the_answers_distance(built, dp_form):
    """`DtwUtils.dtw` returns `min_cost + cb[i + r + 1]` at the first row where that bound reaches bsf (K/utils/DtwUtils.java:324-326),
    and the engines accept a returned value <= eps^2: a row bound that lands EXACTLY on eps^2 makes the window an answer at
    distance eps although its final cell is larger.  On half-integer data with eps = 32 that is the fate of two answers in three
    (found by profiles/stress_r04.sh, seed 9002, against a build of round 4 that carried the minima of one row in V only and
    judged the rest by the final cell: 217 of 273 answers lost).  Every row's minimum has to be looked at -- with abandoning on
    and off, through every register form of the DP."""
    import numpy as np
    import kvmatch
    from kvmatch import datagen
    ctx = kvmatch.Phase2Context(0)
    ctx.set_option(kvmatch.OPT_DP_KERNEL, dp_form)
    seen = at_eps = 0
    for rd, kind, n, m, rho, s, q, qoff, pl, lists, eps_tie, eps_generic in adversarial.cases(datagen, 1, 9002):
        if kind != "halfint":
            continue
        ctx.load_series(s)
        for name, L, R in lists:
            if name == "full":
                continue
            w_off, w_dist, w_st = oracle.rsm_dtw(s, q, L, R, 0, eps_tie, int(rho))
            for abandon in (1, 0):
                ctx.set_option(kvmatch.OPT_EARLY_ABANDON, abandon)
                a = kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps_tie, int(rho))
                assert np.array_equal(a.offsets, w_off) and np.array_equal(a.distances, w_dist), (name, abandon, len(a), len(w_off))
                assert a.cnt_candidate == w_st["cnt_candidate"]
            seen += len(w_off)
            at_eps += int((w_dist == eps_tie).sum())
    ctx.close()
    assert seen > 100 and at_eps > seen // 2      # (the case is what it is meant to be: most answers sit exactly at eps)
