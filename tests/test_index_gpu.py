"""-m gpu: the index build as a whole -- kvm_index_build = IndexBuilder.run for one window length
(K/IndexBuilder.java:186-347) + the bytes of IndexFileOperator.writeAll (K/operator/file/IndexFileOperator.java:127-165) --
against the oracle's restatement (kvo_index_file), byte for byte.  Step 1 runs on the GPU (kvm_index.hip), step 2 / the
codec / the file layout on the library's host side (kvm_index_file.cpp)."""
import numpy as np
import pytest

import java_transliteration as J
import oracle
from test_transliteration import _index_series

pytestmark = pytest.mark.gpu


def test_index_file_equals_the_oracle_byte_for_byte(built, tmp_path):
    import kvmatch
    rng = np.random.default_rng(41)
    with kvmatch.Phase2Context(0) as ctx:
        for trial in range(20):
            kind = trial % 5
            n = int(rng.integers(400, 60_000)) if trial < 15 else int(rng.integers(150_000, 450_000))
            s = _index_series(rng, kind, n)
            ctx.load_series(s)
            for w in (25, 50, 100, 200, 400):
                if n < w:
                    continue
                want, winfo = oracle.index_file(s, w)
                got, info = ctx.index_build(w)
                assert got == want, (trial, n, w, len(got), len(want))
                assert [info[k] for k in ("rows_before", "rows_after", "intervals_before", "intervals_after")] == winfo
        # the file on disk, and read back the way IndexFileOperator does
        path = tmp_path / f"index-{n}-400"
        got2, _ = ctx.index_build(400, path=path)
        assert path.read_bytes() == got2 == got
        rows, triples, offsets = J.read_index_file(got2)
        assert triples[-1][2] == n - 400 + 1
        # errors: no window of that length (the Java code dies on rawStatisticInfo.get(0)); window longer than EPOCH
        ctx.load_series(np.arange(30, dtype=np.float64))
        assert len(ctx.index_build(25)[0]) > 0
        with pytest.raises(kvmatch.KvMatchIllegalArgument):
            ctx.index_build(31)
        with pytest.raises(kvmatch.KvMatchIllegalArgument):
            ctx.index_build(0)


def test_index_file_constant_series_and_long_rows(built):
    """One key for the whole series: a single row of n - w + 1 positions cut every 255 (step 1, :267) -- one line whose
    compact groups close every 255 intervals."""
    import kvmatch
    s = np.full(200_000, 1.25)          # (sums of 1.25 are exact: the mean is 1.25 at every position)
    with kvmatch.Phase2Context(0) as ctx:
        ctx.load_series(s)
        got, info = ctx.index_build(50)
        want, winfo = oracle.index_file(s, 50)
        assert got == want
        assert info["rows_before"] == info["rows_after"] == 1
        rows, triples, _ = J.read_index_file(got)
        assert len(rows) == 1 and rows[0][0] == 1.25
        assert rows[0][1][0] == (1, 255) and rows[0][1][-1][1] == 200_000 - 50 + 1


def test_index_builder_mirror_writes_the_five_files(built, tmp_path):
    """kvmatch.IndexBuilder = K/IndexBuilder.java for the "file" storage type: files/index-{N}-{Wu} for Wu in 25, 50, 100,
    200, 400 (:52-53), each equal to the oracle's image; a window length longer than the series is skipped."""
    import kvmatch
    from kvmatch import datagen
    n = 120_000
    s = datagen.gen_series(n, seed=5)
    with kvmatch.Phase2Context(0) as ctx:
        ctx.load_series(s)
        done = kvmatch.IndexBuilder(ctx, n).build_indexes(tmp_path / "files")
        assert sorted(done) == [25, 50, 100, 200, 400]
        for wu in done:
            assert (tmp_path / "files" / f"index-{n}-{wu}").read_bytes() == oracle.index_file(s, wu)[0]
        ctx.load_series(s[:150])
        assert sorted(kvmatch.IndexBuilder(ctx).build_indexes(tmp_path / "short")) == [25, 50, 100]
        with pytest.raises(kvmatch.KvMatchIllegalArgument):
            kvmatch.IndexBuilder(ctx, n)
