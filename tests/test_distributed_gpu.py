"""-m gpu: the sharded device-resident path above world 1 on hardware -- two ranks sharing the box's single GPU (see
tests/dist_gpu_worker.py for why the group is gloo there), against the unsharded oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_verifier_two_ranks_one_device(built, tmp_path):
    from kvmatch import datagen
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cp = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                         "--master-addr", "127.0.0.1", "--master-port", str(port),
                         os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(tmp_path)],
                        env=env, capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0, cp.stdout[-2000:] + cp.stderr[-4000:]
    n, m, qoff = 400_000, 256, 123_456
    series = datagen.gen_series(n, seed=4242)
    planted = [50_000, 300_000]
    q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.1], [0.0, 0.3], 0.02, seed=5)
    left, right = datagen.gen_intervals(n, m, 60_000, 32.0, must_cover=[qoff] + planted, seed=6)
    want = [oracle.rsm_ed(series, q, left, right, 0, 6.0), oracle.cnsm_ed(series, q, left, right, 0, 4.0, 1.5, 2.0),
            oracle.rsm_dtw(series, q, left, right, 0, 6.0, 12), oracle.cnsm_dtw(series, q, left, right, 0, 4.0, 12, 1.5, 2.0)]
    for variant in range(4):
        w_off, w_dist, w_st = want[variant]
        assert len(w_off) >= 2 and qoff in w_off.tolist()
        for pack_cap in (4096, 2):
            shards = 0
            for rank in range(2):
                z = np.load(tmp_path / f"v{variant}_p{pack_cap}_r{rank}.npz")
                # whole intervals are dealt out (interleaved split), so even the cNSM running sums restart where the
                # unsharded run restarts them: every rank holds the oracle's list bit for bit
                assert np.array_equal(z["offs"], w_off), (variant, pack_cap, rank)
                assert np.array_equal(z["dists"], w_dist), (variant, pack_cap, rank)
                assert int(z["cnt"]) == w_st["cnt_candidate"]
                shards += int(z["shard"])
            assert shards == w_st["cnt_candidate"]
    # the pipelined stream (ShardedStream, two queries in flight per rank): every query of the mixed stream, on every rank
    for i in range(8):
        for rank in range(2):
            z = np.load(tmp_path / f"stream{i}_r{rank}.npz")
            w_off, w_dist, w_st = want[int(z["variant"])]
            assert np.array_equal(z["offs"], w_off) and np.array_equal(z["dists"], w_dist), (i, rank)
            assert int(z["cnt"]) == w_st["cnt_candidate"]
            z = np.load(tmp_path / f"pystream{i}_r{rank}.npz")       # (the same stream through Python slot threads)
            assert np.array_equal(z["offs"], w_off) and np.array_equal(z["dists"], w_dist), (i, rank)

    # a shard-dependent error ends the query on EVERY rank (the job above would have hung in the all-gather otherwise)
    # (third entry on: a shard that one rank cannot even marshal -- left[] and right[] of different lengths -- ends the stream
    # the same way, and the stream is usable afterwards)
    assert (tmp_path / "bad_shard_r0.txt").read_text() == repr(["ShardFailed", "ShardFailed", 3, "ShardFailed", 1])
    assert (tmp_path / "bad_shard_r1.txt").read_text() == repr(["KvMatchIllegalArgument", "KvMatchIllegalArgument", 3,
                                                                "KvMatchIllegalArgument", 1])
    # more answers than the packed block holds (the exact-size second exchange of ShardedStream) -- and a failing shard later
    # in the same stream
    for i in range(3):
        for rank in range(2):
            z = np.load(tmp_path / f"small_cap{i}_r{rank}.npz")
            w_off, w_dist, _ = want[int(z["variant"])]
            assert len(w_off) > 2 and np.array_equal(z["offs"], w_off) and np.array_equal(z["dists"], w_dist), (i, rank)
    assert (tmp_path / "small_cap_bad_shard_r0.txt").read_text() == repr(["ShardFailed", 2])
    assert (tmp_path / "small_cap_bad_shard_r1.txt").read_text() == repr(["KvMatchIllegalArgument", 2])


def test_bench_launched_path_two_ranks_rehearsal(built):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank, ShardedStream with the
    calling thread issuing the all-gathers, max-over-ranks timing, rank 0 printing the one JSON line) -- rehearsed with two
    ranks on this box's single GPU over gloo (KVM_BENCH_REHEARSAL: RCCL refuses two ranks on one device).  Checks the
    plumbing and the line, not a number."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, KVM_BENCH_REHEARSAL="gloo")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for workload, extra in (("rsm_ed", []), ("mixed", ["--queries-per-step", "4"])):
        cp = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                             "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                             "--gpus", "2", "--steps", "2", "--warmup", "1", "--candidates", "30000",   # (the launcher's own
                             # parser takes `--n` for an abbreviation of its options: the series stays at the default 1e8)
                             "--extras", "0", "--cpu-baseline", "0", "--workload", workload] + extra,
                            env=env, capture_output=True, text=True, timeout=900)
        assert cp.returncode == 0, cp.stdout[-2000:] + cp.stderr[-4000:]
        lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, cp.stdout[-2000:]                       # rank 0 alone prints, and prints one line
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["rccl_ranks"] == 0 and "rehearsal" in d and d["scaling"] == "weak"
        assert d["value"] > 0 and d["steps"] == 2 and d["config"]["queries_in_flight"] >= 1
        assert "x2" in d["config"]["parallelism"] and d["config"]["answers"] >= 1   # the merged answers of both shards
