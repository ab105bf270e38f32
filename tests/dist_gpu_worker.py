"""Worker of tests/test_distributed_gpu.py: one rank of a 2-rank job, BOTH ranks on cuda:0 (the test box has one GPU).
RCCL refuses two ranks on one device ("invalid usage"), so the group is gloo -- which moves CUDA tensors too; what is
exercised above world 1 on hardware is everything else of the multi-GPU path: split_intervals, kvm_verify_device (answers
stay in HBM, caller's stream), kvm_pack_answers_device, the fixed-size all_gather_into_tensor of device tensors, the
exact-size fallback exchange, and the merge.  Writes r{rank}.npz per variant into the directory given as argv[1]."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import kvmatch  # noqa: E402
from kvmatch import datagen, distributed as D  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    n, m, qoff = 400_000, 256, 123_456
    series = datagen.gen_series(n, seed=4242)
    planted = [50_000, 300_000]
    q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.1], [0.0, 0.3], 0.02, seed=5)
    left, right = datagen.gen_intervals(n, m, 60_000, 32.0, must_cover=[qoff] + planted, seed=6)
    l, r = D.split_intervals(left, right, world, rank)
    cands = int((r.astype(np.int64) - l + 1).sum())
    args = [dict(epsilon=6.0), dict(epsilon=4.0, alpha=1.5, beta=2.0), dict(epsilon=6.0, rho=12),
            dict(epsilon=4.0, rho=12, alpha=1.5, beta=2.0)]
    with kvmatch.Phase2Context(0) as ctx:
        ctx.load_series(series)
        for variant in range(4):
            for pack_cap in (4096, 2):  # 2 forces the exact-size second exchange
                v = D.ShardedVerifier(ctx, variant, cands, pack_cap=pack_cap)
                offs, dists = v.verify((l, r), 1, q, **args[variant])
                cnt = torch.tensor([v.cnt_candidate], dtype=torch.int64)
                dist.all_reduce(cnt)
                np.savez(os.path.join(out_dir, f"v{variant}_p{pack_cap}_r{rank}.npz"), offs=offs, dists=dists,
                         cnt=int(cnt.item()), shard=cands)
        # the pipelined form: two queries in flight per rank (two contexts over the rank's resident series), the exchange
        # issued by the calling thread alone, in query order; a mixed stream of all four variants, twice
        ss = D.ShardedStream(ctx, 0, cands, in_flight=2)
        queries = [dict(valid_positions=(l, r), last_segment=1, query_data=q, variant=v, **args[v]) for v in (0, 3, 1, 2, 0, 3, 2, 1)]
        res = ss.run(queries)
        cnts = torch.tensor(ss.cnt_candidate, dtype=torch.int64)
        dist.all_reduce(cnts)
        for i, (offs, dists) in enumerate(res):
            np.savez(os.path.join(out_dir, f"stream{i}_r{rank}.npz"), offs=offs, dists=dists, cnt=int(cnts[i].item()),
                     variant=queries[i]["variant"])
        # (the slots' loops are native threads by default; the Python slot threads of rounds 2-3 stay available: same stream)
        ssp = D.ShardedStream(ctx, 0, cands, in_flight=2, native=False)
        for i, (offs, dists) in enumerate(ssp.run(queries)):
            np.savez(os.path.join(out_dir, f"pystream{i}_r{rank}.npz"), offs=offs, dists=dists, variant=queries[i]["variant"])
        ssp.close()
        # one bad shard: rank 1's list ends in an interval beyond the series (readTimeSeries' IllegalArgument, for that rank
        # only).  Nobody may be left waiting in the all-gather: the failing rank takes part with a count of -1, rank 1 raises
        # its own error, rank 0 raises ShardFailed -- from ShardedVerifier.verify and, at the same query, from ShardedStream.run.
        lb, rb = (l, r) if rank == 0 else (np.append(l, n + 10).astype(np.int32), np.append(r, n + 20).astype(np.int32))
        outcome = []
        v = D.ShardedVerifier(ctx, 0, cands + 64)
        try:
            v.verify((lb, rb), 1, q, **args[0])
            outcome.append("no error")
        except Exception as e:  # noqa: BLE001
            outcome.append(type(e).__name__)
        queries = [dict(valid_positions=(l, r), last_segment=1, query_data=q, variant=0, **args[0]),
                   dict(valid_positions=(lb, rb), last_segment=1, query_data=q, variant=0, **args[0]),
                   dict(valid_positions=(l, r), last_segment=1, query_data=q, variant=1, **args[1]),
                   dict(valid_positions=(l, r), last_segment=1, query_data=q, variant=0, **args[0])]
        try:
            ss.run(queries)
            outcome.append("no error")
        except Exception as e:  # noqa: BLE001
            outcome.append(type(e).__name__)
        res = ss.run(queries[:1] + queries[2:])       # and the stream is usable afterwards
        outcome.append(len(res))
        # a shard that cannot even be marshalled on one rank (left[] and right[] of different lengths): the same, collectively
        bad = dict(queries[0], valid_positions=(l, r) if rank == 0 else (l, r[:-1]))
        try:
            ss.run([queries[0], bad, queries[2]])
            outcome.append("no error")
        except Exception as e:  # noqa: BLE001
            outcome.append(type(e).__name__)
        outcome.append(len(ss.run(queries[:1])))
        with open(os.path.join(out_dir, f"bad_shard_r{rank}.txt"), "w") as f:
            f.write(repr(outcome))
        ss.close()
        # the same two things TOGETHER in one stream: more answers than the packed block holds (pack_cap = 2: every query
        # takes the exact-size second exchange) and, further down the same stream, a shard that fails on one rank -- the
        # queries in front of it come back complete on both ranks, the stream ends at the failing query on both
        ss2 = D.ShardedStream(ctx, 0, cands + 64, in_flight=2, pack_cap=2)
        good = [dict(valid_positions=(l, r), last_segment=1, query_data=q, variant=v_, **args[v_]) for v_ in (0, 1, 3)]
        res = ss2.run(good)
        for i, (offs, dists) in enumerate(res):
            np.savez(os.path.join(out_dir, f"small_cap{i}_r{rank}.npz"), offs=offs, dists=dists, variant=good[i]["variant"])
        outcome2 = []
        try:
            ss2.run(good + [dict(valid_positions=(lb, rb), last_segment=1, query_data=q, variant=0, **args[0])] + good[:1])
            outcome2.append("no error")
        except Exception as e:  # noqa: BLE001
            outcome2.append(type(e).__name__)
        outcome2.append(len(ss2.run(good[:2])))          # usable afterwards
        with open(os.path.join(out_dir, f"small_cap_bad_shard_r{rank}.txt"), "w") as f:
            f.write(repr(outcome2))
        ss2.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
