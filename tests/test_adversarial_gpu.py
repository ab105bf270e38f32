"""Adversarial inputs (-m gpu): see tests/adversarial.py.  Every verification must equal the oracle bit for bit --
including the reference's lower-bound ties (DtwUtils.dtw, K/utils/DtwUtils.java:324-326), which integer-valued data with
an integer eps^2 does hit."""
import pytest

import adversarial
import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.strict_sort
def test_adversarial_inputs_all_variants(built):
    import kvmatch
    from kvmatch import datagen
    ctx = kvmatch.Phase2Context(0)
    lines = []
    total, bad, ties = adversarial.run(kvmatch, datagen, oracle, ctx, rounds=1, seed=11, log=lines.append)
    ctx.close()
    assert bad == 0, "\n".join(l for l in lines if l.startswith("MISMATCH"))
    assert total >= 700
    assert ties > 0           # integer-valued data with an integer eps^2 does reach the sequential tie path
