"""Host-side mirror of the reference's four query engines, restricted to the phase-2 seam.

The reference inlines phase 2 ("post-processing") at the tail of each engine's
``query(List<StatisticInfo>, List<Double>, ...)`` (K/QueryEngine.java:338-379,
K/QueryEngineDtw.java:349-468, K/NormQueryEngine.java:432-544, K/NormQueryEngineDtw.java:457-619).  The classes
here keep the reference's names and argument meaning and expose that block as ``verify(...)``: inputs are what
phase 1 hands over (``validPositions`` as (left, right) pairs, ``lastSegment``), outputs are the sorted
``answers`` list and ``cntCandidate``; ``statistics`` slots 2..4 (T_2, #candidates, #answers) are appended the
way K/QueryEngine.java:368-370 does.  Phases 0/1 (index probing) are out of scope and stay in the Java host.

All arithmetic happens in libkvmatch_hip.so through the C ABI; this file only marshals.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

WU0 = 25  # WuList[0], K/QueryEngine.java:44


class StatisticInfo:
    """K/statistic/StatisticInfo.java:43-69 (last / sum / n / min / max accumulator)."""

    def __init__(self):
        self.last, self.sum, self.n = -1.0, 0.0, 0
        self.maximum, self.minimum = 4.9e-324, 1.7976931348623157e308  # Double.MIN_VALUE / MAX_VALUE

    def append(self, value):
        value = float(value)
        self.last = value
        self.sum += value
        self.n += 1
        self.maximum = max(self.maximum, value)
        self.minimum = min(self.minimum, value)

    def get_average(self):
        return self.sum / self.n


class Answers:
    """``List<Pair<Integer, Double>>`` sorted by distance (K/QueryEngine.java:341,373)."""

    def __init__(self, offsets, distances, cnt_candidate, elapsed_ms, stats):
        self.offsets = offsets
        self.distances = distances
        self.cnt_candidate = cnt_candidate
        self.elapsed_ms = elapsed_ms
        self.stats = stats

    def __len__(self):
        return len(self.offsets)

    def pairs(self):
        return list(zip(self.offsets.tolist(), self.distances.tolist()))


def _as_intervals(valid_positions):
    """Accept [(left, right), ...] (any sequence of pairs), a (k, 2) array, or a 2-tuple of one-dimensional NUMPY arrays
    (left[], right[]).  Only numpy arrays select the second form: a tuple of two pairs such as ((1, 5), (10, 20)) is two
    intervals, [1, 5] and [10, 20], as in the reference's List<Interval>."""
    if (isinstance(valid_positions, tuple) and len(valid_positions) == 2
            and all(isinstance(x, np.ndarray) and x.ndim == 1 for x in valid_positions)):
        left, right = valid_positions
        if left.dtype == np.int32 and right.dtype == np.int32 and left.shape == right.shape:
            # the hot path of a query stream: nothing to convert, nothing that could overflow a Java int
            return np.ascontiguousarray(left), np.ascontiguousarray(right)
        left, right = (np.asarray(x, dtype=np.int64) for x in (left, right))
    else:
        arr = np.asarray(valid_positions, dtype=np.int64)
        if arr.size == 0:
            arr = arr.reshape(0, 2)
        if arr.ndim != 2 or arr.shape[1] != 2:
            # (a tuple of two plain lists (left[], right[]) of k != 2 entries ends here instead of being re-read, silently,
            #  as k pairs in the wrong order: that form needs numpy arrays)
            raise _lib.KvMatchIllegalArgument("valid_positions: a sequence of (left, right) pairs, a (k, 2) array, or a 2-tuple "
                                              "of one-dimensional numpy arrays (left[], right[])")
        left, right = arr[:, 0], arr[:, 1]
    if left.shape != right.shape:
        raise _lib.KvMatchIllegalArgument("left/right length mismatch")
    if left.size and (min(left.min(), right.min()) < -2**31 or max(left.max(), right.max()) > 2**31 - 1):
        raise _lib.KvMatchIllegalArgument("interval bounds must fit a Java int")  # (never wrapped silently)
    return np.ascontiguousarray(left, dtype=np.int32), np.ascontiguousarray(right, dtype=np.int32)


def resolve_rho(rho, m, resolved=None):
    """Band half-width in points.  The reference's query(..., int rho) takes points (K/QueryEngineDtw.java:165); its REPL
    reads a double r and resolves `r <= 1 -> floor(r * m), else floor(r)` (:135-140).  Here an integer argument is
    points, a float goes through the prompt rule; `resolved=True/False` forces one reading."""
    if resolved is None:
        resolved = isinstance(rho, (int, np.integer)) and not isinstance(rho, bool)
    if resolved:
        r = int(rho)
        if r < 0 or r > 2**31 - 1:
            raise _lib.KvMatchIllegalArgument("rho must be in [0, 2^31-1]")
        return r
    return int(_lib.load().kvm_resolve_rho(float(rho), int(m)))


def sort_and_merge_intervals(valid_positions):
    """sortAndMergeIntervals (K/QueryEngine.java:664-693): phase 1's `validPositions` in any order -> the sorted, disjoint,
    non-adjacent (left, right) arrays the verify() methods take.  Host only."""
    left, right = _as_intervals(valid_positions)
    ol, orr = np.empty_like(left), np.empty_like(right)
    k = int(_lib.load().kvm_sort_and_merge_intervals(left.ctypes.data, right.ctypes.data, left.size, ol.ctypes.data, orr.ctypes.data))
    if k < 0:
        _lib.check(k, None)
    return ol[:k].copy(), orr[:k].copy()


def cnsm_dtw_order(query):
    """The order NormQueryEngineDtw sums its LB_Keogh bounds in: the permutation JDK 8's Arrays.sort leaves under the
    comparator (int)(|o2| - |o1|) on the z-normalised query (K/NormQueryEngineDtw.java:470-484).  Host only.  Raises
    KvMatchIllegalArgument("Comparison method violates its general contract!") for a query on which that sort throws --
    the reference's query() dies there, and so does NormQueryEngineDtw.verify unless OPT_SORT_VIOLATION is set."""
    q = np.ascontiguousarray(query, dtype=np.float64)
    order = np.empty(q.size, dtype=np.int32)
    rc = _lib.load().kvm_cnsm_dtw_order(q.ctypes.data, q.size, order.ctypes.data)
    if rc:
        _lib.check(rc, None)
    return order


def phase2_time_estimate_ms(variant, disjoint_windows, candidate_offsets, length):
    """The reference's phase-2 cost model with MI355X constants (K/QueryEngine.java:55-57,316-327); variant 0 rsm_ed,
    1 cnsm_ed, 2 rsm_dtw, 3 cnsm_dtw."""
    return float(_lib.load().kvm_phase2_time_estimate_ms(int(variant), int(disjoint_windows), int(candidate_offsets), int(length)))


def phase2_time_estimate_dp_ms(variant, disjoint_windows, candidate_offsets, length, dp_count, num_cu=0):
    """phase2_time_estimate_ms with the number of candidates that reach the banded DP (Answers.stats[STAT_PASS_LB_KEOGH2] of
    a call on the same list): what the DTW engines' time actually follows."""
    return float(_lib.load().kvm_phase2_time_estimate_dp_ms(int(variant), int(disjoint_windows), int(candidate_offsets), int(length),
                                                            int(dp_count), int(num_cu)))


def phase1_should_stop(variant, segment_index, phase1_ms_so_far, disjoint_windows, candidate_offsets, length, last_total_ms):
    """The early-termination test of the phase-1 loop (K/QueryEngine.java:316-327) with a caller-supplied, deterministic
    phase-1 cost instead of the reference's wall clock.  Returns ``(stop, last_total_ms)``: carry the second value into the
    next call (any start value: nothing stops before segment 5 and every earlier call overwrites it)."""
    last = C.c_double(float(last_total_ms))
    rc = _lib.load().kvm_phase1_should_stop(int(variant), int(segment_index), float(phase1_ms_so_far), int(disjoint_windows),
                                            int(candidate_offsets), int(length), C.byref(last))
    if rc < 0:
        raise ValueError("kvm_phase1_should_stop: bad arguments")
    return bool(rc), last.value


class Phase2Context:
    """One HIP device + one staged series; the analogue of the engine constructor
    ``QueryEngine(int n, String storageType)`` (K/QueryEngine.java:70) which opens the TimeSeriesOperator."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.kvm_open(int(device), C.byref(h)))
        self._h = h
        self.device = device
        self._keep = None
        # (this BINDING's own switches, for the parity campaigns of profiles/stress_r06.sh, which run unmodified fuzz scripts with
        # every form of the ED pruner forced: KVM_PY_ED_FORCE=<KVM_OPT_ED_FORCE value>, KVM_PY_IN_FLIGHT=<n>.  The library reads
        # neither)
        if os.environ.get("KVM_PY_ED_FORCE"):
            self.set_option(_lib.OPT_ED_FORCE, int(os.environ["KVM_PY_ED_FORCE"], 0))
        if os.environ.get("KVM_PY_IN_FLIGHT"):
            self.set_option(_lib.OPT_IN_FLIGHT, int(os.environ["KVM_PY_IN_FLIGHT"]))

    # -- TimeSeriesOperator side ---------------------------------------------------------------------
    def load_series(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        _lib.check(self._lib.kvm_load_series_f64(self._h, v.ctypes.data, v.size), self._h)

    def load_series_be64(self, raw_bytes, n):
        buf = np.frombuffer(raw_bytes, dtype=np.uint8)
        if buf.size < n * 8:
            raise _lib.KvMatchIllegalArgument("byte buffer shorter than n*8")
        _lib.check(self._lib.kvm_load_series_be64(self._h, buf.ctypes.data, int(n)), self._h)

    def load_series_file(self, path, n):
        _lib.check(self._lib.kvm_load_series_file(self._h, str(path).encode(), int(n)), self._h)

    def attach_device_series(self, data_ptr, n, keepalive=None):
        self._keep = keepalive
        _lib.check(self._lib.kvm_attach_series_device(self._h, C.c_void_p(int(data_ptr)), int(n)), self._h)

    def share_series(self, owner):
        """Use the series another context on the same device has staged, without copying it (kvm_share_series): one more
        engine over the same data, for a second query in flight.  `owner` must outlive this context's use of it."""
        self._keep = owner
        _lib.check(self._lib.kvm_share_series(self._h, owner._h), self._h)
        return self

    @property
    def n(self):
        return int(self._lib.kvm_series_length(self._h))

    def read_time_series(self, left, length):
        """TimeSeriesOperator.readTimeSeries(left, length), 1-based (K/operator/TimeSeriesOperator.java:38)."""
        out = np.empty(max(int(length), 0), dtype=np.float64)
        _lib.check(self._lib.kvm_read_time_series(self._h, int(left), int(length), out.ctypes.data), self._h)
        return out

    def data_envelope(self, rho, first=1, count=None):
        """(upper, lower) envelope of radius rho of the staged series, DtwUtils.lowerUpperLemire over the whole series
        (K/utils/DtwUtils.java:93-134)."""
        if count is None:
            count = self.n - (first - 1)
        u = np.empty(max(int(count), 0), dtype=np.float64)
        l = np.empty(max(int(count), 0), dtype=np.float64)
        _lib.check(self._lib.kvm_data_envelope(self._h, int(rho), int(first), int(count), u.ctypes.data, l.ctypes.data), self._h)
        return u, l

    def window_means(self, w, first=1, count=None):
        """Sliding means of length-w windows starting at 1-based offsets first .. first+count-1 (index build support,
        cf. K/IndexBuilder.java:232-247)."""
        if count is None:
            count = self.n - w + 1 - (first - 1)
        out = np.empty(max(int(count), 0), dtype=np.float64)
        _lib.check(self._lib.kvm_window_means(self._h, int(w), int(first), int(count), out.ctypes.data), self._h)
        return out

    def index_rows(self, w, grouped=False):
        """Step 1 of the index build for window length w (K/IndexBuilder.java:186-300): the intervals of consecutive
        window positions sharing one row key as (keys, first, last) arrays -- the reference's own running sums and
        MeanIntervalUtils.toRound, bit for bit.  grouped=False: in position order (what the loop appends).
        grouped=True: collected into rows, ordered by (key descending, position) as step 2 walks them (:318); returns
        (keys, first, last, group_keys, group_offsets) with row g = intervals [group_offsets[g], group_offsets[g+1])."""
        import ctypes as C
        h = C.c_void_p()
        build = self._lib.kvm_index_rows_build_grouped if grouped else self._lib.kvm_index_rows_build
        _lib.check(build(self._h, int(w), C.byref(h)), self._h)
        try:
            cnt = int(self._lib.kvm_index_rows_count(h))
            if cnt == 0:
                out = (np.empty(0), np.empty(0, dtype=np.int32), np.empty(0, dtype=np.int32))
                return out + (np.empty(0), np.zeros(1, dtype=np.int32)) if grouped else out
            keys = np.ctypeslib.as_array(self._lib.kvm_index_rows_keys(h), shape=(cnt,)).copy()
            first = np.ctypeslib.as_array(self._lib.kvm_index_rows_first(h), shape=(cnt,)).copy()
            last = np.ctypeslib.as_array(self._lib.kvm_index_rows_last(h), shape=(cnt,)).copy()
            if grouped:
                ng = int(self._lib.kvm_index_rows_groups(h))
                gk = np.ctypeslib.as_array(self._lib.kvm_index_rows_group_keys(h), shape=(ng,)).copy()
                go = np.ctypeslib.as_array(self._lib.kvm_index_rows_group_offsets(h), shape=(ng + 1,)).copy()
        finally:
            self._lib.kvm_index_rows_free(h)
        return (keys, first, last, gk, go) if grouped else (keys, first, last)

    def index_build(self, w, path=None):
        """IndexBuilder.run for window length w, whole (K/IndexBuilder.java:186-347): step 1 on the GPU, step 2 (row
        merging), IndexNode.toBytesCompact and the file layout of IndexFileOperator.writeAll on the host.  Returns
        (bytes of files/index-{N}-{w}, info) with info = dict(rows_before, rows_after, intervals_before, intervals_after,
        step1_ms, step2_ms, encode_ms); `path` also writes the file."""
        import ctypes as C
        h = C.c_void_p()
        _lib.check(self._lib.kvm_index_build(self._h, int(w), C.byref(h)), self._h)
        try:
            size = int(self._lib.kvm_index_file_size(h))
            data = C.string_at(self._lib.kvm_index_file_bytes(h), size)
            names = ("rows_before", "rows_after", "intervals_before", "intervals_after")
            info = {k: int(self._lib.kvm_index_file_info(h, i)) for i, k in enumerate(names)}
            for i, k in ((4, "step1_ms"), (5, "step2_ms"), (6, "encode_ms")):
                info[k] = self._lib.kvm_index_file_info(h, i) / 1e3
            if path is not None:
                rc = self._lib.kvm_index_file_write(h, os.fsencode(path))
                if rc:
                    raise _lib.KvMatchIOError(f"cannot write {path}")
        finally:
            self._lib.kvm_index_file_free(h)
        return data, info

    def measure_device(self, what):
        """What the device sustains (kvm_measure_device): 0 HBM read stream GB/s, 1 fp64 FMA TFLOP/s, 2 fp64 non-FMA cell-mix Tops/s."""
        v = C.c_double()
        _lib.check(self._lib.kvm_measure_device(self._h, int(what), C.byref(v)), self._h)
        return v.value

    def set_option(self, option, value):
        _lib.check(self._lib.kvm_set_option(self._h, int(option), int(value)), self._h)

    def get_option(self, option):
        return int(self._lib.kvm_get_option(self._h, int(option)))

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.kvm_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- result unpacking ---------------------------------------------------------------------------
    def _collect(self, res):
        lib = self._lib
        try:
            cnt = lib.kvm_result_count(res)
            if cnt:
                off = np.ctypeslib.as_array(lib.kvm_result_offsets(res), shape=(cnt,)).copy()
                dist = np.ctypeslib.as_array(lib.kvm_result_distances(res), shape=(cnt,)).copy()
            else:
                off = np.empty(0, dtype=np.int32)
                dist = np.empty(0, dtype=np.float64)
            stats = [int(lib.kvm_result_stat(res, i)) for i in range(8)]
            return Answers(off, dist, int(lib.kvm_result_candidates(res)), float(lib.kvm_result_elapsed_ms(res)),
                           stats)
        finally:
            lib.kvm_result_free(res)


class IndexBuilder:
    """Mirror of K/IndexBuilder.java for the "file" storage type: `IndexBuilder(ctx, n).build_indexes(directory)` writes
    files/index-{n}-{Wu} for the enabled window lengths (WuList with WuEnabled, :52-53 -- 25, 50, 100, 200, 400), each the
    bytes IndexBuilder.SingleIndexBuilder.run + IndexFileOperator.writeAll produce for the series staged in `ctx`
    (step 1 on the GPU, step 2 / codec / layout on the library's host side: kvm_index_build)."""
    WU_LIST = (25, 50, 75, 100, 125, 150, 175, 200, 225, 250, 275, 300, 325, 350, 375, 400)
    WU_ENABLED = (True, True, False, True, False, False, False, True, False, False, False, False, False, False, False, True)

    def __init__(self, ctx, n=None):
        self.ctx = ctx
        self.n = int(ctx.n if n is None else n)
        if self.n != ctx.n:
            raise _lib.KvMatchIllegalArgument("n differs from the length of the staged series")

    def build_indexes(self, directory="files"):
        """Returns {Wu: info} (see Phase2Context.index_build); a window length longer than the series is skipped the way
        the reference's loop logs the exception and carries on (:108-110)."""
        os.makedirs(directory, exist_ok=True)
        done = {}
        for wu, enabled in zip(self.WU_LIST, self.WU_ENABLED):
            if not enabled:
                continue
            try:
                _, info = self.ctx.index_build(wu, path=os.path.join(directory, f"index-{self.n}-{wu}"))
            except _lib.KvMatchIllegalArgument:
                continue
            done[wu] = info
        return done


class QueryStream:
    """Several queries in flight against one resident series: `in_flight` contexts on one device (the first is the
    caller's, the others share its series -- kvm_share_series), each driven by its own host thread, the way a server would
    run several of the reference's (non re-entrant) engine instances over one data file.  While one query's kernels run,
    the next one's host work -- plan sums, query preparation, launches -- and its result read-back proceed, and the tail of
    one pruning kernel overlaps the head of the next.

        with QueryStream(ctx, in_flight=2) as qs:
            results = qs.run([lambda c: QueryEngine(c).verify(iv, 1, q, eps), ...])   # results in job order

    Every job is a callable taking the context it runs on (the ctypes calls release the GIL)."""

    def __init__(self, ctx, in_flight=2):
        self.ctxs = [ctx] + [Phase2Context(ctx.device).share_series(ctx) for _ in range(max(1, int(in_flight)) - 1)]
        for c in self.ctxs[1:]:
            for opt in (_lib.OPT_EARLY_ABANDON, _lib.OPT_LB_CASCADE, _lib.OPT_ED_KERNEL, _lib.OPT_TIMING, _lib.OPT_DP_KERNEL,
                        _lib.OPT_SORT_VIOLATION, _lib.OPT_UCR_LOOPS, _lib.OPT_STRICT_BOUNDS, _lib.OPT_ED_FORCE, _lib.OPT_SERIALIZE_ENQUEUE):
                c.set_option(opt, ctx.get_option(opt))

    def set_option(self, option, value):
        for c in self.ctxs:
            c.set_option(option, value)

    def run(self, jobs):
        import threading
        jobs = list(jobs)
        out = [None] * len(jobs)
        if len(self.ctxs) == 1 or len(jobs) <= 1:
            for i, job in enumerate(jobs):
                out[i] = job(self.ctxs[0])
            return out
        nxt = [0]
        lock = threading.Lock()
        err = []

        def worker(c):
            while not err:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if i >= len(jobs):
                    return
                try:
                    out[i] = jobs[i](c)
                except BaseException as e:  # noqa: BLE001 (re-raised in the caller's thread)
                    err.append(e)

        # (the threads are this interpreter's: the library is told how many contexts they keep busy -- KVM_OPT_IN_FLIGHT)
        busy = min(len(self.ctxs), len(jobs))
        old = [c.get_option(_lib.OPT_IN_FLIGHT) for c in self.ctxs]
        for c in self.ctxs:
            c.set_option(_lib.OPT_IN_FLIGHT, busy)
        try:
            threads = [threading.Thread(target=worker, args=(c,)) for c in self.ctxs[:busy]]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            for c, o in zip(self.ctxs, old):
                c.set_option(_lib.OPT_IN_FLIGHT, o)
        if err:
            raise err[0]
        return out

    def run_jobs(self, jobs, collect=None):
        """The same stream with the per-thread loop in native code (kvm_run_jobs): `jobs` are `Job` descriptors -- calls as
        data -- and every context's host thread is a native one, so the calls are not serialised by this interpreter's lock
        between them (a Python thread holds it for 40 - 80 us per call: marshalling, result objects).  Returns the Answers in
        job order; raises what the first failing job's engine would have raised.  `collect`: indices whose results are
        wanted as Answers objects (default: all); the other results -- complete, sorted, in host memory when the call
        returns -- are released unread and their slots hold None (a throughput driver checks a sample)."""
        jobs = list(jobs)
        n = len(jobs)
        if n == 0:
            return []
        lib = self.ctxs[0]._lib
        arr = (_lib.KvmJob * n)()
        for a, j in zip(arr, jobs):
            a.variant, a.m, a.query, a.left, a.right = j.variant, j.q.size, j.q.ctypes.data, j.left.ctypes.data, j.right.ctypes.data
            a.k, a.shift, a.epsilon, a.rho, a.alpha, a.beta = j.left.size, j.shift, j.epsilon, j.rho, j.alpha, j.beta
        hs = (C.c_void_p * len(self.ctxs))(*[c._h for c in self.ctxs])
        res = (C.c_void_p * n)()
        st = (C.c_int32 * n)()
        rc = lib.kvm_run_jobs(hs, len(self.ctxs), arr, n, res, st)
        if rc:
            for r in res:
                if r:
                    lib.kvm_result_free(r)
            bad = next((i for i in range(n) if st[i]), None)
            # (kvm_run_jobs leaves the first failing job's message in the first context, whichever context ran the job:
            # a context's own text may be that of a failure handled long ago)
            _lib.check(rc, self.ctxs[0]._h)
            raise _lib.KvMatchDeviceError(f"[{rc}] job {bad} failed")
        if collect is None:
            return [self.ctxs[0]._collect(C.c_void_p(r)) for r in res]
        want = set(int(i) % n for i in collect)
        out = [None] * n
        for i, r in enumerate(res):
            if i in want:
                out[i] = self.ctxs[0]._collect(C.c_void_p(r))
            else:
                lib.kvm_result_free(r)
        return out

    def close(self):
        for c in self.ctxs[1:]:
            c.close()
        self.ctxs = self.ctxs[:1]


    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class DeviceGroup:
    """One engine over several GPUs (kvm_group of include/kvmatch.h): the series replicated on every device, the candidate
    list of a call dealt out to the devices, ONE RCCL all-gather of the answers issued from native code -- the host-side
    mirror of `new QueryEngine(n, storageType)` (K/QueryEngine.java:70) for a node of GPUs.

        g = DeviceGroup(devices=[0, 1, ..., 7], in_flight=8)                      # one process, eight GPUs
        g = DeviceGroup(rank=r, world=8, device=local_rank, unique_id=uid)       # one process per GPU (uid from
        g.load_series(series)                                                     #   DeviceGroup.unique_id() on one rank)
        ans = g.verify(0, (left, right), 1, q, 10.0)                              # Answers, as QueryEngine.verify
        res = g.run_jobs([Job(...), ...])                                         # a stream, `in_flight` queries overlapping
    """

    def __init__(self, devices=None, in_flight=1, rccl_always=False, rank=None, world=None, device=None, unique_id=None):
        self._lib = _lib.load()
        self._g = None
        h = C.c_void_p()
        flags = _lib.GROUP_RCCL_ALWAYS if rccl_always else 0
        if rank is None:
            devs = np.ascontiguousarray([0] if devices is None else list(devices), dtype=np.int32)
            rc = self._lib.kvm_open_devices(devs.ctypes.data, devs.size, int(in_flight), flags, C.byref(h))
        else:
            uid = bytes(unique_id) if unique_id is not None else None
            rc = self._lib.kvm_open_rank(int(device if device is not None else 0), int(rank), int(world), uid,
                                         len(uid) if uid is not None else 0, int(in_flight), flags, C.byref(h))
        _lib.check(rc, group=True)
        self._g = h
        self.in_flight = int(in_flight)

    @staticmethod
    def unique_id():
        """The bytes one rank makes and the launcher carries to the others (kvm_group_unique_id)."""
        buf = C.create_string_buffer(_lib.GROUP_ID_BYTES)
        _lib.check(_lib.load().kvm_group_unique_id(buf, _lib.GROUP_ID_BYTES), group=True)
        return buf.raw

    world = property(lambda self: int(self._lib.kvm_group_world(self._g)))
    local_devices = property(lambda self: int(self._lib.kvm_group_local_devices(self._g)))
    rccl_ranks = property(lambda self: int(self._lib.kvm_group_rccl_ranks(self._g)))

    def load_series(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        _lib.check(self._lib.kvm_group_load_series_f64(self._g, v.ctypes.data, v.size), group=self._g)

    def load_series_file(self, path, n):
        _lib.check(self._lib.kvm_group_load_series_file(self._g, os.fsencode(path), int(n)), group=self._g)

    def set_option(self, option, value):
        _lib.check(self._lib.kvm_group_set_option(self._g, int(option), int(value)), group=self._g)

    def configure(self, what, value):
        _lib.check(self._lib.kvm_group_configure(self._g, int(what), int(value)), group=self._g)

    def stats(self):
        """kvm_group_stat of the last stream: where its threads spent their time (microseconds per job)."""
        names = ("jobs", "wait_issued", "prepare", "all_gather", "copy", "wait_block", "merge", "slots_wait", "slots_split", "slots_enqueue", "slots_turn")
        v = [int(self._lib.kvm_group_stat(self._g, i)) for i in range(len(names))]
        jobs = max(1, v[0])
        return {"jobs": v[0], **{nm + "_us_per_job": x / jobs / 1e3 for nm, x in zip(names[1:], v[1:])}}

    def last_kernel_ns(self, local_device=0, slot=0):
        return int(self._lib.kvm_last_kernel_ns(self._lib.kvm_group_ctx(self._g, local_device, slot)))

    def _collect(self, res):
        return Phase2Context._collect(self, res)

    def verify(self, variant, valid_positions, last_segment, query_data, epsilon, rho=0, alpha=0.0, beta=0.0):
        j = Job(variant, valid_positions, last_segment, query_data, epsilon, rho=rho, alpha=alpha, beta=beta)
        res = C.c_void_p()
        rc = self._lib.kvm_verify_sharded(self._g, j.variant, j.q.ctypes.data, j.q.size, j.left.ctypes.data, j.right.ctypes.data,
                                          j.left.size, j.shift, j.epsilon, j.rho, j.alpha, j.beta, C.byref(res))
        _lib.check(rc, group=self._g)
        return self._collect(res)

    def run_jobs(self, jobs, collect=None):
        """kvm_run_sharded_jobs: `jobs` are `Job` descriptors over the WHOLE candidate lists; returns the Answers in job
        order (None where `collect` does not ask for them), raises the first failing job's error."""
        jobs = list(jobs)
        n = len(jobs)
        if n == 0:
            return []
        arr = _job_array(jobs)
        res = (C.c_void_p * n)()
        st = (C.c_int32 * n)()
        rc = self._lib.kvm_run_sharded_jobs(self._g, arr, n, res, st)
        if rc:
            for r in res:
                if r:
                    self._lib.kvm_result_free(r)
            _lib.check(rc, group=self._g)
        want = None if collect is None else set(int(i) % n for i in collect)
        out = [None] * n
        for i, r in enumerate(res):
            if want is None or i in want:
                out[i] = self._collect(C.c_void_p(r))
            else:
                self._lib.kvm_result_free(r)
        return out

    def close(self):
        if self._g is not None:
            self._lib.kvm_group_close(self._g)
            self._g = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def split_intervals(valid_positions, parts, part, mode=0):
    """kvm_split_intervals: shard `part` of `parts` of the list (mode: _lib.SPLIT_AUTO / SPLIT_INTERLEAVED / SPLIT_CONTIGUOUS)."""
    left, right = _as_intervals(valid_positions)
    ol, orr = np.empty(left.size, dtype=np.int32), np.empty(left.size, dtype=np.int32)
    k = _lib.load().kvm_split_intervals(left.ctypes.data, right.ctypes.data, left.size, int(parts), int(part), int(mode),
                                        ol.ctypes.data, orr.ctypes.data)
    if k < 0:
        raise _lib.KvMatchIllegalArgument("bad kvm_split_intervals arguments")
    return ol[:k].copy(), orr[:k].copy()


def _job_array(jobs):
    arr = (_lib.KvmJob * len(jobs))()
    for a, j in zip(arr, jobs):
        a.variant, a.m, a.query, a.left, a.right = j.variant, j.q.size, j.q.ctypes.data, j.left.ctypes.data, j.right.ctypes.data
        a.k, a.shift, a.epsilon, a.rho, a.alpha, a.beta = j.left.size, j.shift, j.epsilon, j.rho, j.alpha, j.beta
    return arr


class Job:
    """One phase-2 call as data, for QueryStream.run_jobs: the arguments of the engine's verify() -- `variant` 0 QueryEngine,
    1 NormQueryEngine, 2 QueryEngineDtw, 3 NormQueryEngineDtw; rho an int (points) or a float (the REPL's fraction rule)."""

    def __init__(self, variant, valid_positions, last_segment, query_data, epsilon, rho=0, alpha=0.0, beta=0.0):
        self.variant = int(variant)
        self.q = np.ascontiguousarray(query_data, dtype=np.float64)
        if self.q.ndim != 1 or self.q.size > 2**31 - 1:
            raise _lib.KvMatchIllegalArgument("query_data must be a one-dimensional sequence")
        self.left, self.right = _as_intervals(valid_positions)
        self.shift = (int(last_segment) - 1) * WU0
        self.epsilon = float(epsilon)
        self.rho = resolve_rho(rho, self.q.size) if self.variant >= 2 else 0
        self.alpha, self.beta = float(alpha), float(beta)


class _EngineBase:
    def __init__(self, ctx):
        self.ctx = ctx

    @staticmethod
    def _shift(last_segment):
        return (int(last_segment) - 1) * WU0  # (lastSegment - 1) * WuList[0], K/QueryEngine.java:346

    @staticmethod
    def _finish(statistics, ans):
        # K/QueryEngine.java:368-370: phase-2 time, #candidates, #answers
        if statistics is not None:
            statistics[2].append(ans.elapsed_ms)
            statistics[3].append(ans.cnt_candidate)
            statistics[4].append(len(ans))
        return ans

    @staticmethod
    def _query(query_data):
        return np.ascontiguousarray(query_data, dtype=np.float64)


class QueryEngine(_EngineBase):
    """RSM-ED (K/QueryEngine.java).  ``verify`` replaces lines 338-363 + the sort of :373."""

    def verify(self, valid_positions, last_segment, query_data, epsilon, statistics=None):
        q = self._query(query_data)
        left, right = _as_intervals(valid_positions)
        res = C.c_void_p()
        c = self.ctx
        _lib.check(c._lib.kvm_verify_rsm_ed(c._h, q.ctypes.data, q.size, left.ctypes.data, right.ctypes.data,
                                            left.size, self._shift(last_segment), float(epsilon), C.byref(res)), c._h)
        return self._finish(statistics, c._collect(res))


class NormQueryEngine(_EngineBase):
    """cNSM-ED (K/NormQueryEngine.java).  ``verify`` replaces lines 432-528."""

    def verify(self, valid_positions, last_segment, query_data, epsilon, alpha, beta, statistics=None):
        q = self._query(query_data)
        left, right = _as_intervals(valid_positions)
        res = C.c_void_p()
        c = self.ctx
        _lib.check(c._lib.kvm_verify_cnsm_ed(c._h, q.ctypes.data, q.size, left.ctypes.data, right.ctypes.data,
                                             left.size, self._shift(last_segment), float(epsilon), float(alpha),
                                             float(beta), C.byref(res)), c._h)
        return self._finish(statistics, c._collect(res))


class QueryEngineDtw(_EngineBase):
    """RSM-DTW (K/QueryEngineDtw.java).  ``verify`` replaces lines 349-452.  ``rho``: an int is the band half-width in
    points (the reference's ``query(..., int rho)``), a float follows the REPL's prompt rule (values <= 1 are a fraction of
    the query length, K/QueryEngineDtw.java:135-140); see ``resolve_rho``."""

    def verify(self, valid_positions, last_segment, query_data, epsilon, rho, statistics=None, resolved=None):
        q = self._query(query_data)
        left, right = _as_intervals(valid_positions)
        res = C.c_void_p()
        c = self.ctx
        r = resolve_rho(rho, q.size, resolved)
        _lib.check(c._lib.kvm_verify_rsm_dtw(c._h, q.ctypes.data, q.size, left.ctypes.data, right.ctypes.data,
                                             left.size, self._shift(last_segment), float(epsilon), r,
                                             C.byref(res)), c._h)
        return self._finish(statistics, c._collect(res))


class NormQueryEngineDtw(_EngineBase):
    """cNSM-DTW (K/NormQueryEngineDtw.java).  ``verify`` replaces lines 457-603."""

    def verify(self, valid_positions, last_segment, query_data, epsilon, rho, alpha, beta, statistics=None, resolved=None):
        """``rho``: int = points, float = the REPL's fraction rule (``resolve_rho``); ``resolved`` forces one reading."""
        q = self._query(query_data)
        left, right = _as_intervals(valid_positions)
        res = C.c_void_p()
        c = self.ctx
        r = resolve_rho(rho, q.size, resolved)
        _lib.check(c._lib.kvm_verify_cnsm_dtw(c._h, q.ctypes.data, q.size, left.ctypes.data, right.ctypes.data,
                                              left.size, self._shift(last_segment), float(epsilon), r, float(alpha),
                                              float(beta), C.byref(res)), c._h)
        return self._finish(statistics, c._collect(res))
