"""Multi-GPU host layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The candidate set shards embarrassingly: the series is replicated on every GPU and the interval list is
dealt out to the ranks balanced by candidate count -- WHOLE intervals, interleaved (split_intervals: within every round
of `world` consecutive intervals the longest goes to the rank that is lightest so far; contiguous slices that cut inside
an interval only for lists of fewer than 8 intervals per rank).  Neighbouring intervals land on different GPUs (the
expensive candidates of a DTW query cluster around the matches), and because no interval is cut a sharded cNSM run
restarts the reference's running sums exactly where the unsharded one does: bit-identical answers.  Each rank verifies
its shard with the answers left in HBM (kvm_verify_device), and ONE all-gather moves them: a fixed-size block of
[count | first pack_cap (offset, distance) pairs] per rank as fp64 pairs (an int32 offset is exact in fp64); only if some
rank holds more than pack_cap answers does a second, exact-size exchange follow (gather_answers).  The payload is tiny
(16 B per answer), so the collective is latency-bound; xGMI bandwidth is irrelevant to it.  A rank whose shard fails takes
part with a count of -1, so that nobody is left waiting.  Every rank ends with the full answer list, sorted by
(distance, offset) exactly as the single-GPU path sorts (K/QueryEngine.java:373).
"""
import numpy as np


def split_intervals(left, right, world, rank, mode="auto"):
    """Shard `rank` of `world` of the candidate list, balanced by candidate count.  Returns (left, right) int32 arrays;
    every candidate offset appears in exactly one shard and cntCandidate summed over shards equals the unsharded one.

    mode "interleaved": whole intervals are dealt out, always to the rank holding the fewest candidates so far among
        the next `world` intervals (a round of the deal), so neighbouring intervals land on different ranks.  The
        expensive candidates of a DTW query cluster around the matches (a few adjacent intervals carry all the DPs),
        so this is what balances the GPUs; and because no interval is cut, a sharded cNSM run restarts its running sums
        exactly where the unsharded one does (bit-identical answers).
    mode "contiguous": contiguous slices of the candidate sequence, cut inside an interval where a boundary falls
        (the pieces of a cut interval are adjacent, not merged); exact balance even for a handful of long intervals.
    mode "auto" (default): interleaved when there are at least 8 intervals per rank, else contiguous."""
    left = np.asarray(left, dtype=np.int64)
    right = np.asarray(right, dtype=np.int64)
    counts = np.maximum(right - left + 1, 0)
    k = len(left)
    if mode == "auto":
        mode = "interleaved" if k >= 8 * world else "contiguous"
    if mode == "interleaved":
        if world == 1:
            keep = counts > 0
            return left[keep].astype(np.int32), right[keep].astype(np.int32)
        # rounds of `world` consecutive intervals; inside a round the longest interval goes to the rank that is
        # lightest so far.  Vectorised: ranks sorted by load (ascending) meet the round's intervals sorted by length
        # (descending).  The loads stay within one interval length of each other.
        owner = np.empty(k, dtype=np.int64)
        load = np.zeros(world, dtype=np.int64)
        for base in range(0, k, world):
            c = counts[base:base + world]
            by_len = np.argsort(-c, kind="stable")
            by_load = np.argsort(load, kind="stable")[:len(c)]
            owner[base + by_len] = by_load
            np.add.at(load, by_load, c[by_len])
        mine = (owner == rank) & (counts > 0)
        return left[mine].astype(np.int32), right[mine].astype(np.int32)
    if mode != "contiguous":
        raise ValueError("mode must be 'auto', 'interleaved' or 'contiguous'")
    total = int(counts.sum())
    lo = (total * rank) // world        # first global candidate index of this shard
    hi = (total * (rank + 1)) // world  # one past the last
    if hi <= lo:
        return np.empty(0, dtype=np.int32), np.empty(0, dtype=np.int32)
    cum = np.concatenate(([0], np.cumsum(counts)))
    first = int(np.searchsorted(cum, lo, side="right") - 1)
    last = int(np.searchsorted(cum, hi, side="left") - 1)
    l = left[first:last + 1].copy()
    r = right[first:last + 1].copy()
    l[0] = left[first] + (lo - cum[first])
    r[-1] = left[last] + (hi - cum[last]) - 1
    keep = r >= l
    return l[keep].astype(np.int32), r[keep].astype(np.int32)


def sort_answers(offsets, distances):
    """(distance, offset) order == the reference's stable sort by distance of an offset-ascending list."""
    order = np.lexsort((offsets, distances))
    return offsets[order], distances[order]


def gather_answers(offsets, distances, group=None, device=None):
    """All-gather variable-length (offset, distance) lists; returns the merged, sorted arrays on every rank.

    `offsets`/`distances` may be numpy arrays (CPU / gloo) or torch tensors on `device` (RCCL)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if isinstance(offsets, np.ndarray):
        dev = torch.device(device) if device is not None else torch.device("cpu")
        packed = torch.empty((len(offsets), 2), dtype=torch.float64)
        packed[:, 0] = torch.from_numpy(np.asarray(offsets, dtype=np.float64))
        packed[:, 1] = torch.from_numpy(np.asarray(distances, dtype=np.float64))
        packed = packed.to(dev)
    else:
        dev = offsets.device
        packed = torch.stack((offsets.to(torch.float64), distances.to(torch.float64)), dim=1)
    n_local = torch.tensor([packed.shape[0]], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, n_local, group=group)
    counts_h = counts.cpu().numpy()
    mx = int(counts_h.max()) if world else 0
    if mx == 0:
        return np.empty(0, dtype=np.int32), np.empty(0, dtype=np.float64)
    pad = torch.zeros((mx, 2), dtype=torch.float64, device=dev)
    pad[: packed.shape[0]] = packed
    allbuf = torch.empty((world * mx, 2), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(allbuf, pad, group=group)
    allh = allbuf.cpu().numpy().reshape(world, mx, 2)
    offs = np.concatenate([allh[r, : counts_h[r], 0] for r in range(world)]).astype(np.int32)
    dists = np.concatenate([allh[r, : counts_h[r], 1] for r in range(world)])
    return sort_answers(offs, dists)


def sharded_verify(engine, valid_positions, last_segment, query_data, *params, group=None):
    """Verify this rank's shard with `engine.verify(...)` (HIP path) and all-gather the answers.

    Returns (offsets, distances, cnt_candidate) where cnt_candidate is the global count."""
    import torch
    import torch.distributed as dist

    left, right = valid_positions
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    l, r = split_intervals(left, right, world, rank)
    ans = engine.verify((l, r), last_segment, query_data, *params)
    dev = f"cuda:{engine.ctx.device}" if dist.get_backend(group) == "nccl" else "cpu"
    offs, dists = gather_answers(ans.offsets, ans.distances, group=group, device=dev)
    cnt = torch.tensor([ans.cnt_candidate], dtype=torch.int64, device=dev)
    dist.all_reduce(cnt, group=group)
    return offs, dists, int(cnt.item())


class ShardFailed(RuntimeError):
    """Another rank raised while verifying its shard of a query (every rank learns it from the exchange and raises: nobody is
    left waiting in a collective)."""


class ShardedVerifier:
    """Device-resident sharded verification: kernels write survivors straight into torch tensors on this rank's GPU
    (kvm_verify_device, launched on torch's current stream), the survivors are packed on the device and exchanged
    with ONE fixed-size all-gather (count + up to `pack_cap` pairs per rank; a second, exact-size exchange only if
    some rank exceeds `pack_cap`), then copied to the host once and sorted.

    variant: 0 rsm_ed, 1 cnsm_ed, 2 rsm_dtw, 3 cnsm_dtw (the `variant` argument of kvm_verify_device)."""

    def __init__(self, ctx, variant, max_candidates, group=None, pack_cap=4096):
        import torch
        import torch.distributed as dist
        self.ctx, self.variant, self.group = ctx, int(variant), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.dev = torch.device(f"cuda:{ctx.device}")
        self.cap = int(max_candidates) + 1
        self.pack_cap = int(pack_cap)
        self.out_off = torch.empty(self.cap, dtype=torch.int32, device=self.dev)
        self.out_dist = torch.empty(self.cap, dtype=torch.float64, device=self.dev)
        self.count = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.packed = torch.zeros((self.pack_cap + 1, 2), dtype=torch.float64, device=self.dev)
        self.gathered = torch.empty((self.world * (self.pack_cap + 1), 2), dtype=torch.float64, device=self.dev)
        self.host = torch.empty((self.world * (self.pack_cap + 1), 2), dtype=torch.float64).pin_memory()
        self.cnt_candidate = 0

    def launch(self, valid_positions, last_segment, query_data, epsilon, rho=0, alpha=1.0, beta=0.0, resolved=None,
               variant=None):
        """Enqueue the verification of this rank's shard and the pack kernel on torch's current stream; nothing is
        exchanged or copied to the host (the packed block [count | first pack_cap pairs] stays in self.packed).
        `variant` overrides the verifier's own for this query (a mixed query stream)."""
        variant = self.variant if variant is None else int(variant)
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib
        from .engine import WU0, _as_intervals, resolve_rho
        left, right = _as_intervals(valid_positions)
        q = np.ascontiguousarray(query_data, dtype=np.float64)
        c = self.ctx
        r = resolve_rho(rho, q.size, resolved) if variant >= 2 else 0
        cnt_cand = C.c_int64(0)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.check(c._lib.kvm_verify_device(
            c._h, variant, q.ctypes.data, q.size, left.ctypes.data, right.ctypes.data, left.size,
            (int(last_segment) - 1) * WU0, float(epsilon), int(r), float(alpha), float(beta),
            C.c_void_p(self.out_off.data_ptr()), C.c_void_p(self.out_dist.data_ptr()), self.cap,
            C.c_void_p(self.count.data_ptr()), C.byref(cnt_cand), C.c_void_p(stream)), c._h)
        self.cnt_candidate = int(cnt_cand.value)
        # pack [count | first pack_cap pairs] on the device (one kernel), exchange, one copy to pinned host memory
        k = self.pack_cap
        _lib.check(c._lib.kvm_pack_answers_device(c._h, C.c_void_p(self.out_off.data_ptr()),
                                                  C.c_void_p(self.out_dist.data_ptr()), C.c_void_p(self.count.data_ptr()),
                                                  k, C.c_void_p(self.packed.data_ptr()), C.c_void_p(stream)), c._h)

    def verify(self, valid_positions, last_segment, query_data, epsilon, rho=0, alpha=1.0, beta=0.0, resolved=None):
        import torch
        import torch.distributed as dist
        # A shard-dependent error (a range error of THIS rank's intervals, an unsorted shard, out of memory) must not leave the
        # other ranks waiting in the all-gather: the failing rank still takes part, with a count of -1 in its block, and
        # every rank raises once the exchange is through.
        local_err = None
        try:
            self.launch(valid_positions, last_segment, query_data, epsilon, rho, alpha, beta, resolved)
        except Exception as e:  # noqa: BLE001
            if self.world == 1:
                raise
            local_err = e
            self.packed[0, 0] = -1.0
        k = self.pack_cap
        if self.world > 1:
            dist.all_gather_into_tensor(self.gathered, self.packed, group=self.group)
            self.host.copy_(self.gathered, non_blocking=True)
        else:
            self.host[: k + 1].copy_(self.packed, non_blocking=True)
        torch.cuda.current_stream(self.dev).synchronize()
        host = self.host.numpy().reshape(self.world, k + 1, 2)
        counts = host[:, 0, 0].astype(np.int64)
        if local_err is not None:
            raise local_err
        if counts.min() < 0:
            raise ShardFailed(f"rank(s) {np.nonzero(counts < 0)[0].tolist()} failed on their shard of this query")
        if counts.max() <= k:
            offs = np.concatenate([host[w, 1:1 + counts[w], 0] for w in range(self.world)]).astype(np.int32)
            dists = np.concatenate([host[w, 1:1 + counts[w], 1] for w in range(self.world)])
            return sort_answers(offs, dists)
        # rare: more answers than the packed buffer holds -> exact-size exchange of the full lists
        n_local = int(self.count.item())
        return gather_answers(self.out_off[:n_local], self.out_dist[:n_local], group=self.group) \
            if self.world > 1 else sort_answers(self.out_off[:n_local].cpu().numpy(),
                                                self.out_dist[:n_local].cpu().numpy())


class ShardedStream:
    """A stream of sharded queries with `in_flight` of them overlapping on every rank.

    Each slot has its own context (sharing the rank's resident series), its own HIP stream and its own device buffers;
    slot threads run the device-resident verification and the pack kernel (kvm_verify_device, kvm_pack_answers_device) of
    the queries dealt to them -- native threads inside the library (kvm_run_device_jobs) since round 4: between the calls of a
    Python slot thread the interpreter lock is held for 40 - 80 us, which bounds a rank's shards once they are shorter than
    that (strong scaling); `native=False` keeps the Python threads.  The exchange is NOT done by those threads: collectives must be issued in the same order on
    every rank, so the calling thread alone performs the all-gathers, in query order, on its own stream, as each query's
    packed block becomes ready -- the all-gather and the host copy of query i overlap the kernels of query i+1.

        ss = ShardedStream(ctx, variant, max_candidates, in_flight=2)
        results = ss.run([dict(valid_positions=(l, r), last_segment=1, query_data=q, epsilon=10.0), ...])   # or kvmatch.Job objects
        # -> [(offsets, distances), ...] in query order, the merged sorted answer list on every rank
    """

    def __init__(self, ctx, variant, max_candidates, in_flight=2, group=None, pack_cap=4096, native=True):
        import torch
        import torch.distributed as dist
        from .engine import Phase2Context
        self.group, self.variant, self.native = group, int(variant), bool(native)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.dev = torch.device(f"cuda:{ctx.device}")
        self.pack_cap = int(pack_cap)
        self.slots = []
        for i in range(max(1, int(in_flight))):
            c = ctx if i == 0 else Phase2Context(ctx.device).share_series(ctx)
            if i:
                from . import _lib
                for opt in (_lib.OPT_EARLY_ABANDON, _lib.OPT_LB_CASCADE, _lib.OPT_ED_KERNEL, _lib.OPT_TIMING, _lib.OPT_DP_KERNEL,
                            _lib.OPT_SORT_VIOLATION, _lib.OPT_UCR_LOOPS, _lib.OPT_STRICT_BOUNDS, _lib.OPT_ED_FORCE, _lib.OPT_SERIALIZE_ENQUEUE):
                    c.set_option(opt, ctx.get_option(opt))
            stream = torch.cuda.Stream(device=self.dev)
            v = ShardedVerifier(c, variant, max_candidates, group=group, pack_cap=pack_cap)
            ready = torch.cuda.Event()
            ready.record(stream)     # (creates the HIP event: the native loop records it behind each of the slot's jobs)
            self.slots.append(dict(ctx=c, v=v, stream=stream, ready=ready))
        self.comm_stream = torch.cuda.Stream(device=self.dev)
        k = self.pack_cap + 1
        self.gathered = [torch.empty((self.world * k, 2), dtype=torch.float64, device=self.dev) for _ in self.slots]
        self.host = [torch.empty((self.world * k, 2), dtype=torch.float64).pin_memory() for _ in self.slots]
        self.cnt_candidate = []

    def set_option(self, option, value):
        for s in self.slots:
            s["ctx"].set_option(option, value)

    def kernel_ns(self):
        return [int(s["ctx"]._lib.kvm_last_kernel_ns(s["ctx"]._h)) for s in self.slots]

    def run(self, queries):
        queries = list(queries)
        if self.native and queries:
            return self._run_native(queries)
        return self._run_python(queries)

    def _run_native(self, queries):
        """The slots' loops in native threads (kvm_run_device_jobs, started from one helper thread that sits in the C call with
        the interpreter lock released); this thread polls `produced`, runs the exchange of each query in order and hands the
        slot back through `consumed`."""
        import ctypes as C
        import threading
        import time
        import torch
        import torch.distributed as dist
        from . import _lib
        from .engine import Job, resolve_rho
        nq, ns = len(queries), len(self.slots)
        lib = self.slots[0]["ctx"]._lib
        # A query that cannot even be marshalled on THIS rank (its shard of the list is malformed, say) must not leave the other
        # ranks waiting in its all-gather: it becomes a job the library refuses (no query), i.e. a failed shard like any other --
        # the slot takes part with the failure mark and this rank raises its own error when the stream reaches that query.
        jobs_py, marshal_err = [], {}
        for qi, q in enumerate(queries):
            if isinstance(q, Job):     # (marshalled before: a caller that repeats its queries keeps the descriptors)
                jobs_py.append(q)
                continue
            try:
                var = self.variant if q.get("variant") is None else int(q["variant"])
                rho = resolve_rho(q.get("rho", 0), len(q["query_data"]), q.get("resolved")) if var >= 2 else 0
                jobs_py.append(Job(var, q["valid_positions"], q["last_segment"], q["query_data"], q["epsilon"], rho=int(rho),
                                   alpha=q.get("alpha", 1.0), beta=q.get("beta", 0.0)))
            except Exception as e:  # noqa: BLE001
                if self.world == 1:
                    raise
                jobs_py.append(None)
                marshal_err[qi] = e
        arr = (_lib.KvmJob * nq)()
        for a, j in zip(arr, jobs_py):
            if j is None:
                a.variant, a.m, a.k = 0, 0, 0       # (query = NULL, m = 0: KVM_ERR_ILLEGAL_ARGUMENT before anything is touched)
                continue
            a.variant, a.m, a.query, a.left, a.right = j.variant, j.q.size, j.q.ctypes.data, j.left.ctypes.data, j.right.ctypes.data
            a.k, a.shift, a.epsilon, a.rho, a.alpha, a.beta = j.left.size, j.shift, j.epsilon, j.rho, j.alpha, j.beta
        slots = (_lib.KvmDeviceSlot * ns)()
        for t, s in enumerate(self.slots):
            v = s["v"]
            slots[t].ctx, slots[t].stream = s["ctx"]._h, C.c_void_p(s["stream"].cuda_stream)
            slots[t].d_off, slots[t].d_dist, slots[t].cap = v.out_off.data_ptr(), v.out_dist.data_ptr(), v.cap
            slots[t].d_count, slots[t].d_packed, slots[t].pack_cap = v.count.data_ptr(), v.packed.data_ptr(), v.pack_cap
            slots[t].ready = s["ready"].cuda_event
        produced, consumed = (C.c_int32 * nq)(), (C.c_int32 * nq)()
        stop, status, cnts = (C.c_int32 * 1)(), (C.c_int32 * nq)(), (C.c_int64 * nq)()
        rc_box = []
        helper = threading.Thread(target=lambda: rc_box.append(lib.kvm_run_device_jobs(slots, ns, arr, nq, produced, consumed, stop, cnts, status)))
        helper.start()
        out, err = [], None
        k = self.pack_cap
        if "issued" not in self.slots[0]:
            for s in self.slots:
                s["issued"] = torch.cuda.Event()

        def issue(qi):
            """query qi's exchange and host copy, enqueued on the exchange stream behind the slot's kernels; nothing waits here"""
            spins = 0
            while not produced[qi]:
                spins += 1
                if spins < 400:       # (a few tens of microseconds of polling first: the shortest sleep this interpreter can take
                    continue          #  is 50 - 60 us, as long as an early-abandoning query's whole chain)
                if not helper.is_alive() and not produced[qi]:
                    raise _lib.KvMatchDeviceError("the slots' native loop ended before query %d was enqueued" % qi)
                time.sleep(1e-5)
            si = qi % ns
            s = self.slots[si]
            if status[qi] and self.world == 1:
                _lib.check(int(status[qi]), s["ctx"]._h)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(s["ready"])     # (recorded by the native loop behind this query's kernels)
                if self.world > 1:
                    dist.all_gather_into_tensor(self.gathered[si], s["v"].packed, group=self.group)
                    self.host[si].copy_(self.gathered[si], non_blocking=True)
                else:
                    self.host[si][: k + 1].copy_(s["v"].packed, non_blocking=True)
                s["issued"].record(self.comm_stream)

        def finish(qi):
            """wait for query qi's block, merge it, hand the slot back; returns an error (the stream ends here) or None"""
            si = qi % ns
            s = self.slots[si]
            s["issued"].synchronize()
            host = self.host[si].numpy().reshape(self.world, k + 1, 2)
            counts = host[:, 0, 0].astype(np.int64)
            if counts.min() < 0:   # some rank failed on its shard of this query: every rank sees it here and stops
                if qi in marshal_err:
                    return marshal_err[qi]
                if status[qi]:
                    try:
                        _lib.check(int(status[qi]), s["ctx"]._h)
                    except Exception as e:  # noqa: BLE001
                        return e
                return ShardFailed(f"rank(s) {np.nonzero(counts < 0)[0].tolist()} failed on their shard of query {qi}")
            if counts.max() <= k:
                offs = np.concatenate([host[w, 1:1 + counts[w], 0] for w in range(self.world)]).astype(np.int32)
                dists = np.concatenate([host[w, 1:1 + counts[w], 1] for w in range(self.world)])
                out.append(sort_answers(offs, dists))
            else:  # rare: more answers than the packed block holds -> exact-size exchange of the slot's full lists
                with torch.cuda.stream(self.comm_stream):
                    n_local = int(s["v"].count.item())
                    vo, vd = s["v"].out_off[:n_local], s["v"].out_dist[:n_local]
                    out.append(gather_answers(vo, vd, group=self.group) if self.world > 1
                               else sort_answers(vo.cpu().numpy(), vd.cpu().numpy()))
            consumed[qi] = 1
            return None

        # The exchange of query i + 1 is enqueued BEFORE the block of query i is waited for (one query of lag, when there are at
        # least two slots): the host's wait, merge and sort of one query overlap the all-gather of the next.  The order of the
        # collectives -- gather i + 1, then the rare exact-size exchange of query i -- depends on the counts alone, which every
        # rank sees alike.
        lag = 1 if ns > 1 else 0
        try:
            for qi in range(nq + lag):
                if qi < nq:
                    issue(qi)
                if qi - lag >= 0:
                    err = finish(qi - lag)
                    if err is not None:
                        break
        finally:
            stop[0] = 1
            helper.join()
            torch.cuda.synchronize(self.dev)   # (slots that ran ahead of an early end: their kernels read the job arrays above)
        if err is not None:
            raise err
        self.cnt_candidate = [int(c) for c in cnts]
        return out

    def _run_python(self, queries):
        import threading
        import torch
        import torch.distributed as dist
        nq, ns = len(queries), len(self.slots)
        produced = [threading.Event() for _ in range(nq)]
        consumed = [threading.Event() for _ in range(nq)]
        err = []
        cnts = [0] * nq
        stop = threading.Event()   # set by the calling thread once the stream has ended (early) on every rank

        def slot_thread(si):
            s = self.slots[si]
            try:
                failed = False
                for qi in range(si, nq, ns):
                    if qi - ns >= 0:
                        consumed[qi - ns].wait()      # the slot's buffers are free again
                    if stop.is_set():
                        return
                    with torch.cuda.stream(s["stream"]):
                        if not failed:
                            try:
                                s["v"].launch(**queries[qi])  # verify + pack on the slot's stream, no synchronisation
                            except Exception as e:  # noqa: BLE001
                                # a shard-dependent error: the other ranks are (or will be) in this query's all-gather -- take
                                # part in it with a count of -1, so that every rank ends the stream at this query and raises
                                if self.world == 1:
                                    raise
                                err.append((qi, e))
                                failed = True
                        if failed:
                            s["v"].packed[0, 0] = -1.0
                        s["ready"].record(s["stream"])
                    cnts[qi] = s["v"].cnt_candidate
                    produced[qi].set()
            except BaseException as e:  # noqa: BLE001
                err.append((-1, e))
                for ev in produced:
                    ev.set()

        threads = [threading.Thread(target=slot_thread, args=(si,)) for si in range(ns)]
        for t in threads:
            t.start()
        out = []
        k = self.pack_cap
        try:
            for qi in range(nq):
                produced[qi].wait()
                if err and (self.world == 1 or any(q < 0 for q, _ in err)):
                    break
                si = qi % ns
                s = self.slots[si]
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(s["ready"])
                    if self.world > 1:
                        dist.all_gather_into_tensor(self.gathered[si], s["v"].packed, group=self.group)
                        self.host[si].copy_(self.gathered[si], non_blocking=True)
                    else:
                        self.host[si][: k + 1].copy_(s["v"].packed, non_blocking=True)
                    self.comm_stream.synchronize()
                    host = self.host[si].numpy().reshape(self.world, k + 1, 2)
                    counts = host[:, 0, 0].astype(np.int64)
                    if counts.min() < 0:   # some rank failed on its shard of this query: every rank sees it here and stops
                        if not any(q == qi for q, _ in err):
                            err.append((qi, ShardFailed(f"rank(s) {np.nonzero(counts < 0)[0].tolist()} failed on their shard of "
                                                        f"query {qi}")))
                        break
                    if counts.max() <= k:
                        offs = np.concatenate([host[w, 1:1 + counts[w], 0] for w in range(self.world)]).astype(np.int32)
                        dists = np.concatenate([host[w, 1:1 + counts[w], 1] for w in range(self.world)])
                        out.append(sort_answers(offs, dists))
                    else:  # rare: more answers than the packed block holds -> exact-size exchange of the slot's full lists
                        n_local = int(s["v"].count.item())
                        vo, vd = s["v"].out_off[:n_local], s["v"].out_dist[:n_local]
                        out.append(gather_answers(vo, vd, group=self.group) if self.world > 1
                                   else sort_answers(vo.cpu().numpy(), vd.cpu().numpy()))
                consumed[qi].set()
        finally:
            stop.set()
            for ev in consumed:
                ev.set()
            for t in threads:
                t.join()
        if err:
            raise sorted(err, key=lambda qe: (qe[0] < 0, qe[0]))[0][1]   # the earliest query's error (this rank's own if it has one)
        self.cnt_candidate = cnts
        return out

    def close(self):
        for s in self.slots[1:]:
            s["ctx"].close()
        self.slots = self.slots[:1]
