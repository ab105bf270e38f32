"""Child process of tests/test_parity_gpu.py::test_streams_with_the_enqueue_lock_forced: a mixed stream of all four variants
through kvm_run_jobs (four contexts) and through the device-resident slots (kvm_run_device_jobs) with the library's enqueue
lock ON (KVM_SERIALIZE_ENQUEUE=1 in the environment: the lock is decided once per process) -- answers equal the oracle's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "kv-match_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import kvmatch  # noqa: E402
import oracle  # noqa: E402
from kvmatch import datagen, distributed  # noqa: E402
from conftest import sortable_query  # noqa: E402


def same(got_off, got_dist, want):
    return np.array_equal(got_off, want[0]) and np.array_equal(got_dist, want[1])


def main():
    assert os.environ.get("KVM_SERIALIZE_ENQUEUE") == "1"
    n, m = 300_000, 512
    s = datagen.gen_series(n, seed=4711)
    qoff = sortable_query(s, m, 100_000)
    q = datagen.plant_copies(s, qoff, m, [30_000, 220_000], [1.0, 1.1], [0.0, 0.3], 0.02, seed=3)
    L, R = datagen.gen_intervals(n, m, 40_000, 48.0, must_cover=[qoff, 30_000, 220_000], seed=3)
    rho = kvmatch.resolve_rho(0.05, m)
    want = [oracle.rsm_ed(s, q, L, R, 0, 6.0), oracle.cnsm_ed(s, q, L, R, 0, 3.0, 1.5, 2.0),
            oracle.rsm_dtw(s, q, L, R, 0, 6.0, rho), oracle.cnsm_dtw(s, q, L, R, 0, 3.0, rho, 1.5, 2.0)]
    ctx = kvmatch.Phase2Context(0)
    assert ctx.get_option(kvmatch.OPT_SERIALIZE_ENQUEUE) == 1     # (the default of a process started with the variable set)
    ctx.load_series(s)
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
    jobs = [kvmatch.Job(0, (L, R), 1, q, 6.0), kvmatch.Job(1, (L, R), 1, q, 3.0, alpha=1.5, beta=2.0),
            kvmatch.Job(2, (L, R), 1, q, 6.0, rho=rho), kvmatch.Job(3, (L, R), 1, q, 3.0, rho=rho, alpha=1.5, beta=2.0)] * 6
    qs = kvmatch.QueryStream(ctx, in_flight=4)
    for i, got in enumerate(qs.run_jobs(jobs)):
        assert same(got.offsets, got.distances, want[i % 4]), ("run_jobs", i)
    qs.close()
    ss = distributed.ShardedStream(ctx, 0, int((R.astype(np.int64) - L + 1).sum()), in_flight=3)
    for i, (o, d) in enumerate(ss.run(jobs[:12])):
        assert same(o, d, want[i % 4]), ("sharded stream", i)
    ss.close()
    ctx.close()
    print("enqueue lock: 24 + 12 queries equal the oracle")


if __name__ == "__main__":
    main()
