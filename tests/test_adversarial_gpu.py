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
