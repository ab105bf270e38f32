"""-m gpu: the native group entry points (include/kvmatch.h: kvm_group) on hardware -- one device, the exchange through REAL
RCCL (ncclCommInitAll / ncclCommInitRank + ncclAllGather issued by the library itself, KVM_GROUP_RCCL_ALWAYS), against the
unsharded oracle bit for bit: offsets, order, every distance, cntCandidate.  More ranks than one need more GPUs than this
box has (RCCL refuses two ranks on one device); the protocol above one rank runs on the CPU in tests/test_sharded.py."""
import numpy as np
import pytest

import oracle
from conftest import sortable_query

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case(built):
    import kvmatch
    from kvmatch import datagen
    n, m = 400_000, 256
    series = datagen.gen_series(n, seed=4242)
    qoff = sortable_query(series, m, 123_456)
    planted = [50_000, 300_000]
    q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.1], [0.0, 0.3], 0.02, seed=5)
    left, right = datagen.gen_intervals(n, m, 60_000, 32.0, must_cover=[qoff] + planted, seed=6)
    want = [oracle.rsm_ed(series, q, left, right, 0, 6.0), oracle.cnsm_ed(series, q, left, right, 0, 4.0, 1.5, 2.0),
            oracle.rsm_dtw(series, q, left, right, 0, 6.0, 12), oracle.cnsm_dtw(series, q, left, right, 0, 4.0, 12, 1.5, 2.0)]
    for w in want:
        assert len(w[0]) >= 2 and qoff in w[0].tolist()
    args = [dict(epsilon=6.0), dict(epsilon=4.0, alpha=1.5, beta=2.0), dict(epsilon=6.0, rho=12), dict(epsilon=4.0, rho=12, alpha=1.5, beta=2.0)]
    return kvmatch, series, q, (left, right), want, args


def same(ans, want):
    o_off, o_dist, o_st = want
    return (np.array_equal(ans.offsets, o_off) and np.array_equal(ans.distances, o_dist) and ans.cnt_candidate == o_st["cnt_candidate"])


@pytest.mark.parametrize("rccl", [True, False])
def test_group_of_one_device_every_variant_single_calls_and_a_stream(case, rccl):
    kvmatch, series, q, iv, want, args = case
    with kvmatch.DeviceGroup(devices=[0], in_flight=3, rccl_always=rccl) as g:
        assert g.world == 1 and g.local_devices == 1 and g.rccl_ranks == (1 if rccl else 0)
        with pytest.raises(kvmatch.KvMatchDeviceError, match="no series"):
            g.verify(0, iv, 1, q, 6.0)
        g.load_series(series)
        for v in range(4):
            assert same(g.verify(v, iv, 1, q, **args[v]), want[v]), v           # kvm_verify_sharded: one call per query
        jobs = [kvmatch.Job(i % 4, iv, 1, q, **args[i % 4]) for i in range(14)]
        res = g.run_jobs(jobs)                                                   # kvm_run_sharded_jobs: 3 in flight
        assert all(same(r, want[i % 4]) for i, r in enumerate(res))
        res = g.run_jobs(jobs, collect=[13])
        assert res[:13] == [None] * 13 and same(res[13], want[1])
        # the reference's early abandoning off, contiguous slices (cut inside intervals): the same answers
        g.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
        g.configure(kvmatch._lib.GROUP_SPLIT_MODE, kvmatch._lib.SPLIT_CONTIGUOUS)
        assert same(g.verify(0, iv, 1, q, **args[0]), want[0]) and same(g.verify(2, iv, 1, q, **args[2]), want[2])


def test_group_overflow_of_the_packed_block_and_a_failing_job(case):
    kvmatch, series, q, iv, want, args = case
    with kvmatch.DeviceGroup(devices=[0], in_flight=2, rccl_always=True) as g:
        g.load_series(series)
        g.configure(kvmatch._lib.GROUP_PACK_CAP, 1)       # every query with two answers or more takes the exact-size second exchange
        jobs = [kvmatch.Job(i % 4, iv, 1, q, **args[i % 4]) for i in range(7)]
        assert all(same(r, want[i % 4]) for i, r in enumerate(g.run_jobs(jobs)))
        # a job the reference would die on (an interval wholly beyond the series: readTimeSeries' range check,
        # K/operator/file/TimeSeriesFileOperator.java:55-57) ends the stream with the reference's exception ...
        n = len(series)
        bad = (np.append(iv[0], n + 10).astype(np.int32), np.append(iv[1], n + 50).astype(np.int32))
        with pytest.raises(oracle.OracleError):
            oracle.rsm_ed(series, q, bad[0], bad[1], 0, 6.0)
        jobs[4] = kvmatch.Job(0, bad, 1, q, 6.0)
        with pytest.raises(kvmatch.KvMatchIllegalArgument, match="left"):
            g.run_jobs(jobs)
        with pytest.raises(kvmatch.KvMatchIllegalArgument, match="left"):
            g.verify(0, bad, 1, q, 6.0)
        # ... and the group goes on
        jobs[4] = kvmatch.Job(0, iv, 1, q, **args[0])
        assert all(same(r, want[i % 4]) for i, r in enumerate(g.run_jobs(jobs)))


def test_group_as_one_rank_of_a_launched_job(case):
    """kvm_open_rank: what one process per GPU calls (bench.py under torch.distributed.run) -- here a world of one rank, the
    communicator made from a kvm_group_unique_id (ncclCommInitRank)."""
    kvmatch, series, q, iv, want, args = case
    uid = kvmatch.DeviceGroup.unique_id()
    assert len(uid) == 128
    with kvmatch.DeviceGroup(rank=0, world=1, device=0, unique_id=uid, in_flight=2, rccl_always=True) as g:
        assert g.world == 1 and g.rccl_ranks == 1
        g.load_series(series)
        jobs = [kvmatch.Job(i % 4, iv, 1, q, **args[i % 4]) for i in range(6)]
        assert all(same(r, want[i % 4]) for i, r in enumerate(g.run_jobs(jobs)))
    with pytest.raises(kvmatch.KvMatchIllegalArgument):
        kvmatch.DeviceGroup(rank=0, world=2, device=0, unique_id=None)
    with pytest.raises(kvmatch.KvMatchIllegalArgument):
        kvmatch.DeviceGroup(devices=[0, 0])
