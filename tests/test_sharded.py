"""CPU tests of the native group layer (kv-match_amd/csrc/kvm_sharded.cpp; include/kvmatch.h: kvm_group): the split rule
against its Python statement on 10^4 random lists, and the whole layer -- slot loops, the exchange protocol around one
all-gather per query, merge, overflow, failing shards, failing allocations -- compiled as it is over stand-ins for the
device layer and the verification path (tests/harness/sharded_harness.cpp), under ThreadSanitizer and under ASan + UBSan,
with 1 .. 8 devices in one process and 2 .. 3 ranks of one group each (the N > 1 path, without a GPU)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _random_list(rng, it):
    k = int(rng.choice([0, 1, 2, 3, 5, 8, 17, 64, 65, 200, 1000])) if it % 50 else int(rng.integers(0, 5000))
    kind = it % 4
    if kind == 0:      # phase-1 shaped: sorted, disjoint, geometric lengths
        gaps, lens = rng.geometric(0.01, size=k), rng.geometric(1 / 64, size=k)
        left = (np.cumsum(gaps + np.concatenate(([0], lens[:-1]))) + 1) if k else np.zeros(0)
        right = left + lens - 1
    elif kind == 1:    # empty and one-point intervals, overlaps
        left = np.sort(rng.integers(1, 10**6, size=k))
        right = left + rng.integers(-3, 4, size=k)
    elif kind == 2:    # a few huge ones
        left = np.sort(rng.integers(1, 2**30, size=k))
        right = left + rng.integers(0, 2**20, size=k)
    else:              # equal lengths: ties in every round of the deal
        left = np.arange(k) * 100 + 1
        right = left + int(rng.integers(0, 50))
    return left.astype(np.int32), right.astype(np.int32)


def test_kvm_split_intervals_equals_the_python_rule_on_10000_random_lists(built):
    """kvm_split_intervals (C++) == kvmatch.distributed.split_intervals (numpy), every mode, on 10^4 seeded lists; and what
    the group layer relies on: empty intervals belong to no shard, interleaved shards hold whole intervals."""
    import kvmatch
    from kvmatch import distributed
    rng = np.random.default_rng(7)
    modes = {"auto": 0, "interleaved": 1, "contiguous": 2}
    for it in range(10000):
        left, right = _random_list(rng, it)
        world = int(rng.choice([1, 2, 3, 4, 7, 8, 16]))
        rank = int(rng.integers(0, world))
        mname = ("auto", "interleaved", "contiguous")[it % 3]
        a = distributed.split_intervals(left, right, world, rank, mode=mname)
        b = kvmatch.split_intervals((left, right), world, rank, modes[mname])
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (it, world, rank, mname)
        assert np.all(b[1] >= b[0])
        if mname == "interleaved":
            assert set(zip(b[0].tolist(), b[1].tolist())) <= set(zip(left.tolist(), right.tolist()))


def test_group_layer_under_thread_and_address_sanitizers(tmp_path):
    """kvm_sharded.cpp as it is, over the stand-ins of tests/harness/sharded_harness.cpp: in-process groups of 1 / 2 / 4 / 8
    devices with 1 and 3 queries in flight, groups of 2 and 3 ranks (one host thread each) on one id, the exact-size second
    exchange, a shard that fails (its own status on its rank, KVM_ERR_SHARD_FAILED on the others, every rank at the same
    job, the groups usable afterwards), refused and malformed jobs, device allocations that fail at open and in a stream,
    and everything given back at the end."""
    src = [os.path.join(ROOT, "tests", "harness", "sharded_harness.cpp"), os.path.join(ROOT, "kv-match_amd", "csrc", "kvm_sharded.cpp")]
    for san in ("thread", "address,undefined"):
        exe = tmp_path / ("sharded_harness_" + san.split(",")[0])
        r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsanitize=" + san,
                            "-fno-sanitize-recover=all", "-pthread", "-o", str(exe)] + src, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "\nFAIL" not in r.stdout and r.stdout.count("ok  ") == 18, r.stdout + r.stderr[-3000:]
        assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and \
            "runtime error" not in r.stderr, r.stderr[-3000:]
