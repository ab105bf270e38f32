"""Index-free full scans on the GPU: host mirrors of the reference's UCR-suite style executors
(``U/`` = src/main/java/cn/edu/fudan/dsm/kvmatch/experiments/ucr/), which its experiments use as ground truth for the
cNSM engines (K/experiments/NormQueryTestGroupBySelectivity.java:115-126, NormQueryDtwTestGroupBySelectivity.java:116).

"Every offset is a candidate" is the phase-2 path with a candidate list that covers the series; the
only thing that distinguishes the executors from the engines is where the running sums ex / ex2 (and, for DTW, the
data envelope) restart, and that is decided by how the list is cut:

* ``UcrDtwQueryExecutor`` works in buffers of EPOCH = 100000 points carrying M - 1 (U/UcrDtwQueryExecutor.java:97,
  168-214): identical, bit for bit, to NormQueryEngineDtw on intervals of EPOCH - M + 1 offsets (each is read as a
  chunk of its own, K/NormQueryEngineDtw.java:492-524).  Offsets are reported 0-based, as the reference does (:283).
* ``UcrEdQueryExecutor`` keeps ONE pair of running sums over the whole series (U/UcrEdQueryExecutor.java:139-171):
  identical to NormQueryEngine on the single interval [1, N - M + 1].  An exact replay of that chain is sequential by
  nature (one lane, ~10-18 ns a point: 1-2 s at N = 1e8), so ``segment=`` cuts the list into intervals of that many
  offsets instead -- the engine's own semantics (sums restart per interval), equal to the reference's single chain up
  to the last bits of mean / std (i.e. same answers except at a rounding knife edge) and three orders of magnitude
  faster on a long series.
"""
import numpy as np

from . import _lib
from .engine import Answers, NormQueryEngine, NormQueryEngineDtw

EPOCH = 100000  # U/UcrDtwQueryExecutor.java:97


def segments(first, last, length):
    """[first, last] cut into consecutive intervals of `length` offsets (the last one shorter)."""
    if last < first:
        return np.empty(0, dtype=np.int32), np.empty(0, dtype=np.int32)
    left = np.arange(first, last + 1, length, dtype=np.int64)
    right = np.minimum(left + length - 1, last)
    return left.astype(np.int32), right.astype(np.int32)


class _Executor:
    def __init__(self, ctx, N, M, queryData, Epsilon, Alpha, Beta):
        self.ctx, self.N, self.M = ctx, int(N), int(M)
        self.queryData = np.ascontiguousarray(queryData, dtype=np.float64)
        self.Epsilon, self.Alpha, self.Beta = float(Epsilon), float(Alpha), float(Beta)
        self.answers = None
        if self.queryData.size != self.M:
            raise _lib.KvMatchIllegalArgument(f"query has {self.queryData.size} points, M = {self.M}")
        if self.N > ctx.n or self.N < self.M:
            raise _lib.KvMatchIllegalArgument(f"N = {self.N} outside [M, series length {ctx.n}]")


class UcrEdQueryExecutor(_Executor):
    """U/UcrEdQueryExecutor.java.  ``run()`` returns the number of answers (:174); ``answers`` holds them sorted by
    distance, 1-based offsets (:161).  ``segment=None`` replays the reference's single running-sum chain.
    The executor's distance() stops on ``sum < bsf`` where the engines' loops stop on ``dist <= bsf``: the difference is
    reproduced for Epsilon^2 == 0 (see run()); a partial sum that lands EXACTLY on Epsilon^2 before the last term -- which
    the executor would report at distance Epsilon -- is not (a known difference; rare, but integer-valued data
    produces it: tests/golden/ucr/ucr_ed_partial_sum_tie.npz, test_ucr_ed_partial_sum_tie_is_the_known_difference)."""

    def __init__(self, ctx, N, M, queryData, Epsilon, Alpha, Beta, segment=None):
        super().__init__(ctx, N, M, queryData, Epsilon, Alpha, Beta)
        self.segment = None if segment is None else int(segment)
        if self.segment is not None and self.segment < 1:
            raise _lib.KvMatchIllegalArgument("segment must be positive")

    def run(self):
        last = self.N - self.M + 1
        lr = segments(1, last, last if self.segment is None else self.segment)
        if self.Epsilon * self.Epsilon == 0.0:
            # The executor's own distance() abandons on `sum < bsf` (U/UcrEdQueryExecutor.java:93) and accepts on
            # `dist <= bsf` (:165): with bsf = Epsilon * Epsilon == 0 the loop never runs, dist stays 0, and EVERY window
            # that passes the (Alpha, Beta) test is reported at distance 0 -- the engines, whose loop condition is
            # `dist <= bsf` (K/NormQueryEngine.java:513), compute the first term instead.  Reproduced here: the windows
            # that pass the filter (they all are answers for an infinite threshold), in scan order, at distance 0.
            a = NormQueryEngine(self.ctx).verify(lr, 1, self.queryData, float("inf"), self.Alpha, self.Beta)
            offs = np.sort(a.offsets)
            self.answers = Answers(offs, np.zeros(offs.size), a.cnt_candidate, a.elapsed_ms, a.stats)
            return len(self.answers)
        self.answers = NormQueryEngine(self.ctx).verify(lr, 1, self.queryData, self.Epsilon, self.Alpha, self.Beta)
        return len(self.answers)


class UcrDtwQueryExecutor(_Executor):
    """U/UcrDtwQueryExecutor.java.  ``run()`` returns the number of answers (:314); ``answers`` holds them sorted by
    distance, 0-BASED offsets (:283).  Known difference: the executor never clears cb1 / cb2 between windows
    (:157-161), so at an exact LB_Keogh tie it can reject a window the engines -- and this mirror -- report at distance
    Epsilon (tests/golden/ucr/ucr_dtw_stale_cb_tie.npz)."""

    def __init__(self, ctx, N, M, queryData, Epsilon, Rho, Alpha, Beta):
        super().__init__(ctx, N, M, queryData, Epsilon, Alpha, Beta)
        self.Rho = int(Rho)
        if self.M > EPOCH:
            raise _lib.KvMatchIllegalArgument(f"M = {self.M} exceeds EPOCH = {EPOCH}")

    def run(self):
        lr = segments(1, self.N - self.M + 1, EPOCH - self.M + 1)
        a = NormQueryEngineDtw(self.ctx).verify(lr, 1, self.queryData, self.Epsilon, self.Rho, self.Alpha, self.Beta,
                                                resolved=True)
        self.answers = Answers(a.offsets - 1, a.distances, a.cnt_candidate, a.elapsed_ms, a.stats)
        return len(self.answers)
