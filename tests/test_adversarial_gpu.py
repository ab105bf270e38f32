"""Adversarial inputs (-m gpu): see tests/adversarial.py.  Every verification must equal the oracle bit for bit; the only
tolerated divergence is the reference's lower-bound tie (DtwUtils.dtw, SURVEY 8a fact iii), recognised explicitly."""
import pytest

import adversarial
import oracle

pytestmark = pytest.mark.gpu


def test_adversarial_inputs_all_variants(built):
    import kvmatch
    from kvmatch import datagen
    ctx = kvmatch.Phase2Context(0)
    lines = []
    total, bad, quirk = adversarial.run(kvmatch, datagen, oracle, ctx, rounds=1, seed=11, log=lines.append)
    ctx.close()
    assert bad == 0, "\n".join(l for l in lines if l.startswith("MISMATCH"))
    assert total >= 700
    assert quirk > 0          # integer-valued data with an integer eps^2 does hit the reference's tie
