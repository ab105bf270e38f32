// kvm_verify.hip -- the phase-2 driver: run_variant (plan -> query block -> pruners -> exact pass, per variant), the sequential tie
// path of the DTW engines, result assembly, and the kvm_verify_* / kvm_verify_device / kvm_result_* entry points
#include "kvm_api_internal.h"
#include "kvm_jsort.h"
#include <mutex>

// KVM_OPT_SERIALIZE_ENQUEUE: the ENQUEUE phases of a context's verification calls (plan, copies, launches: everything but the
// waits for the stream) hold one process-wide lock: the kernels of different contexts still overlap on the GPU, the host
// threads take turns issuing them.  For hosts that run under a tool that hooks the runtime's kernel dispatches (rocprofv3
// --kernel-trace): several host threads enqueueing at the same moment take that tool's library down -- a SIGSEGV in a copy
// inside its dispatch callback, on whichever thread is launching (profiles/r05_prof_native_crash.log; never seen without the
// tool).  The library does not look for such a tool: the host says so (bench.py sets the option on its contexts when it finds
// the profiler's environment, and reports it), or KVM_SERIALIZE_ENQUEUE=1 makes it the default of the contexts a process opens.
struct kvm_enqueue_lock {
    std::unique_lock<std::mutex> lk;
    const bool on;
    explicit kvm_enqueue_lock(const kvm_ctx *ctx) : on(ctx && ctx->opt_serialize_enqueue) { again(); }
    void release() {
        if (lk.owns_lock()) lk.unlock();
    }
    void again() {
        if (on && !lk.owns_lock()) lk = std::unique_lock<std::mutex>(mu());
    }
    static std::mutex &mu() {
        static std::mutex m;
        return m;
    }
};

// ---- the shared verification driver ---------------------------------------------------------------------

struct verify_out {
    int32_t *d_off;
    double *d_dist;
    int64_t cap;
    bool own;  // true: answers go to the context's own buffers, sized by run_variant once the plan knows #windows
    // cNSM-DTW with KVM_OPT_UCR_LOOPS: what the sequential tie path needs to reconstruct the executor's never-cleared cb1 / cb2
    kvm_replay_ws rws = {};
    const kvm_tile *tiles = nullptr;
    int64_t n_tiles = 0, windows = 0;
    bool ucr_tables_ready = false;
};

enum { V_RSM_ED = 0, V_CNSM_ED = 1, V_RSM_DTW = 2, V_CNSM_DTW = 3 };

// The envelope is built on the stream of the call that needs it (its consumers run on that stream).  A cached envelope
// may have been built on ANOTHER stream by an earlier kvm_verify_device call that returned without synchronising: the
// event recorded behind the build orders this call's stream after it.
static int ensure_envelope(kvm_ctx *ctx, int32_t rho, hipStream_t stream) {
    // An earlier kvm_verify_device call may still be running on ANOTHER stream and reading the envelope arrays (run_variant
    // has made its own stream wait already and cleared the flag; kvm_data_envelope comes here directly): order this stream
    // behind it on the device before anything below can rebuild them.
    if (ctx->done_pending && ctx->done_stream != stream) HIPCHK(ctx, hipStreamWaitEvent(stream, ctx->ev_done, 0));
    if (ctx->env_rho == rho && ctx->d_env_u) {
        if (ctx->ev_env && ctx->env_stream != stream) HIPCHK(ctx, hipStreamWaitEvent(stream, ctx->ev_env, 0));
        return KVM_OK;
    }
    if (!ctx->d_env_u) {
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_env_u, (size_t)ctx->n * 8));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_env_l, (size_t)ctx->n * 8));
    }
    ctx->env_rho = -1;
    // (a rebuild for another rho overwrites arrays the previous call's kernels may still be reading on another stream: this
    // stream has been ordered behind that call's ev_done -- recorded on every exit of kvm_verify_device -- above or in run_variant)
    HIPCHK(ctx, kvm_launch_envelope(ctx->d_series, ctx->n, rho, ctx->d_env_u, ctx->d_env_l, stream));
    if (!ctx->ev_env) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_env, hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(ctx->ev_env, stream));
    ctx->env_stream = stream;
    ctx->env_rho = rho;
    return KVM_OK;
}

// Upper / lower envelope of radius rho of the resident series, points first .. first + count - 1 (1-based): what
// DtwUtils.lowerUpperLemire (K/utils/DtwUtils.java:93-134) returns for the whole series fetched as one chunk.  The DTW
// engines use it for every point that is at least rho away from both ends of its fetched chunk.
extern "C" int kvm_data_envelope(kvm_ctx *ctx, int32_t rho, int32_t first, int64_t count, double *upper, double *lower) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!ctx->d_series) return kvm_fail(ctx, KVM_ERR_NO_SERIES, "no series loaded");
    if (rho < 0 || first < 1 || count < 0 || (int64_t)first + count - 1 > ctx->n || !upper || !lower)
        return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "bad kvm_data_envelope arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_envelope(ctx, rho, ctx->stream);
    if (rc) return rc;
    if (count > 0) {
        HIPCHK(ctx, hipMemcpyAsync(upper, ctx->d_env_u + (first - 1), (size_t)count * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(lower, ctx->d_env_l + (first - 1), (size_t)count * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return KVM_OK;
    });
}

// Runs the kernels of one variant; survivors land (unsorted) in out.d_off/out.d_dist, counters stay on the
// device in ctx->b_counters.  `stream` is the stream to launch on.
static int run_variant(kvm_ctx *ctx, int variant, const double *query, int32_t m, const int32_t *left,
                       const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho, double alpha,
                       double beta, plan &pl, verify_out &out, hipStream_t stream, int64_t *kernel_ns,
                       kvm_query_dev &dtw_qd, kvm_dtw_aux &dtw_ax) {
    if (!ctx->d_series) return kvm_fail(ctx, KVM_ERR_NO_SERIES, "no series loaded");
    if (!query || m < 1 || k < 0 || (k > 0 && (!left || !right)))
        return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "bad query/interval arguments");
    if (m > ctx->n) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "query longer than the series");
    if (!(epsilon >= 0)) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "epsilon must be >= 0");
    const bool cnsm = variant == V_CNSM_ED || variant == V_CNSM_DTW;
    const bool dtw = variant == V_RSM_DTW || variant == V_CNSM_DTW;
    // (the REPLs refuse `alpha <= 0.0` (K/NormQueryEngine.java:139-142); a NaN passes that test, and in the engine neither
    //  `std / stdQ <= alpha` nor `>= 1 / alpha` holds for it (:511): no answers -- the guarded filters keep every candidate
    //  (a NaN compares false) and the exact (alpha, beta) test, which is the reference's own, drops them all)
    if (cnsm && alpha <= 0) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "alpha must be > 0");
    if (dtw && rho < 0) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "rho must be >= 0");
    // ---- query-side constants (host, sequential, exactly the reference's arithmetic) ----
    std::vector<double> q(query, query + m), qu, ql;
    kvm_query_dev qd;
    std::memset(&qd, 0, sizeof qd);
    qd.m = m;
    qd.rho = rho;
    qd.eps2 = epsilon * epsilon;
    qd.normalised = cnsm ? 1 : 0;
    qd.early_abandon = (int32_t)ctx->opt_early_abandon;
    qd.alpha = alpha;
    qd.beta = beta;
    if (cnsm) {
        // K/NormQueryEngine.java:192-198 (phase 0) and :438-441 (z-normalisation)
        double ex = 0, ex2 = 0;
        for (int i = 0; i < m; i++) {
            ex += query[i];
            ex2 += query[i] * query[i];
        }
        qd.meanQ = ex / m;
        qd.stdQ = std::sqrt(ex2 / m - qd.meanQ * qd.meanQ);
        for (int i = 0; i < m; i++) q[i] = (query[i] - qd.meanQ) / qd.stdQ;
    }
    // DtwUtils.lowerUpperLemire(Q, length, rho, l, u) (K/QueryEngineDtw.java:362, K/NormQueryEngineDtw.java:469) stores
    // `u[i - r - 1]` from i = len on (K/utils/DtwUtils.java:81-90): with rho >= m that is ArrayIndexOutOfBoundsException
    // before anything has been read -- and before the sort below.  KVM_OPT_STRICT_BOUNDS = 0 computes the natural extension.
    if (dtw && ctx->opt_strict_bounds && m <= rho) {
        char msg[32];
        snprintf(msg, sizeof msg, "%d", (int)(m - rho - 1));
        return kvm_fail(ctx, KVM_ERR_INDEX_OUT_OF_BOUNDS, msg);
    }
    if (dtw) kvm_query_envelope(q, rho, qu, ql);
    // The reference's summation order of the LB_Keogh loops.  RSM-DTW: the reorder is commented out, order = identity
    // (K/QueryEngineDtw.java:364-383).  cNSM-DTW: Arrays.sort with the comparator (int)(|o2| - |o1|)
    // (K/NormQueryEngineDtw.java:475-478) -- inconsistent, so the permutation is the one JDK 8's TimSort leaves
    // (kvm_jsort.h), and the sort may fail the way Java throws: query() then ends in that IllegalArgumentException before
    // it has read anything, so this comes first here too.  KVM_OPT_SORT_VIOLATION = 1 answers such a query anyway,
    // summing in plain index order (the answers differ from a successful sort's only at exact ties with eps^2).
    std::vector<int32_t> ord;
    if (dtw) {
        ord.resize((size_t)m);
        for (int i = 0; i < m; i++) ord[(size_t)i] = i;
        if (cnsm && !kvm_jsort::cnsm_dtw_order(q.data(), m, ord.data())) {
            if (!ctx->opt_sort_violation)
                return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "Comparison method violates its general contract!");
            for (int i = 0; i < m; i++) ord[(size_t)i] = i;
        }
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // Scratch buffers and the pinned staging area are reused by every call: wait until the previous call's device
    // work has drained (kvm_verify_device returns without synchronising the caller's stream).
    if (ctx->done_pending) {
        // the previous kvm_verify_device call may still be reading the device scratch on its stream: this call's stream waits
        // for it on the DEVICE (a call on the same stream is ordered behind it anyway)
        if (ctx->done_stream != stream) HIPCHK(ctx, hipStreamWaitEvent(stream, ctx->ev_done, 0));
        ctx->done_pending = false;
    }
    if (ctx->staged_pending) {
        HIPCHK(ctx, hipEventSynchronize(ctx->ev_staged));
        ctx->staged_pending = false;
    }
    // ED pruning kernel: 1 = wave-per-candidate (the cross-check), 3 = LDS-tiled correlation form; 0 = auto = 3.  (2, the direct
    // LDS-tiled form, and 4, lane-granular tiles staged by LDS-DMA, were removed in rounds 3 and 4: the latter stayed 2 - 3 %
    // behind the default -- HISTORY.md.)
    const int ed_kernel = ctx->opt_ed_kernel != 0 ? (int)ctx->opt_ed_kernel : 3;
    auto tp_ = std::chrono::steady_clock::now();
    // 32-candidate tiles for the LDS-tiled kernels (ED pruners, fused DTW lower-bound stage), 64 otherwise
    // (DTW cascade: 64-candidate tiles first, trimmed and cut to <= 32 on the device by kvm_retile_dtw_kernel)
    // cNSM-ED with the dot pruner: uniform 64-candidate tiles first, trimmed and classified on the device by the filter;
    // RSM-ED with early abandoning: the same, trimmed by the head of the window (kvm_retile_rsm_kernel)
    const bool retile_rsm = !cnsm && !dtw && ed_kernel == 3 && ctx->opt_early_abandon;
    const bool retile = (cnsm && !dtw && ed_kernel == 3) || retile_rsm;
    // host-planned list for the correlation-form pruner (RSM-ED, every point evaluated): the class A / B / C tiling
    // (kvm_ed_dot.hip)
    const int32_t tile_max = retile ? KVM_TILE_MAX : (!dtw && ed_kernel == 3) ? KVM_TILE_TILED : KVM_TILE_MAX;
    const bool hybrid = !dtw && ed_kernel == 3 && !retile;
    // query block: q | (qu | ql for DTW) | pad || (qc | pad | q_pref8 | qk = the three phase arrays of qc, for ED) || (zQ
    // sorted | order: cNSM-ED exact pass); the runs follow it in the pinned staging buffer, where build_plan writes them directly
    // (the phase arrays serve the 16-per-lane class A, which runs only where every point is evaluated: an abandoning call --
    // bound by its launch chain and by the host's 50 us per call -- neither builds nor uploads their 3 (m / 2 + 1024) doubles)
    const bool want_qk = !dtw && !ctx->opt_early_abandon;
    const int nchunks8 = (m + ED8_JC - 1) / ED8_JC, nk = want_qk ? kvm_qk_len(m) : 0;
    const size_t qmain = (size_t)m * 8 * (dtw ? 3 : 1) + KVM_SERIES_PAD * sizeof(double);
    const size_t qpref_off = (size_t)m * 8 + KVM_SERIES_PAD * sizeof(double), qk_off = qpref_off + ((size_t)nchunks8 + 1) * 16;
    const size_t qdot = dtw ? 0 : qk_off + (size_t)3 * nk * 8;
    const size_t qexact_off = qmain + qdot;
    // DTW: the reference's summation order of the LB_Keogh loops and the query / its envelope in that order
    // (uo | lo | qo | order) behind the main block
    const size_t qbytes = (qexact_off + ((cnsm && !dtw) ? (size_t)m * 12 : 0) + (dtw ? (size_t)m * 28 : 0) + 15) / 16 * 16;
    // the interval list follows the query block (left | right, each padded to 16 B; cNSM-DTW: + the chunk bounds per
    // interval); the runs built from it on the device follow that
    const bool chunks = cnsm && dtw;
    const size_t lr_half = ((size_t)(k > 0 ? k : 0) * 4 + 15) / 16 * 16, lr_bytes = (chunks ? 4 : 2) * lr_half;
    int rc = kvm_ensure_pinned(ctx, qbytes + lr_bytes + 64);
    if (rc) return rc;
    char *hp = (char *)ctx->h_pinned;
    pl.left = (int32_t *)(hp + qbytes);
    pl.right = (int32_t *)(hp + qbytes + lr_half);
    if (chunks) {
        pl.chunk_lo = (int32_t *)(hp + qbytes + 2 * lr_half);
        pl.chunk_hi = (int32_t *)(hp + qbytes + 3 * lr_half);
    }
    rc = kvm_build_plan(ctx, cnsm, m, left, right, k, shift, tile_max, hybrid, pl, (dtw && ctx->opt_strict_bounds) ? rho : -1);
    if (rc) return rc;
    // Class-C tiles (two <= 16-candidate remainders per row, kvm_ed_dot.hip) fill the lanes better but a class-C unit
    // lasts 1.3 x as long as a class-B unit: always worth it when the list needs more than one round of the chip's wave
    // slots (throughput); below that the launcher's model of the launch decides (kvm_ed_dot_choose_classes).  Decided
    // here, from the totals; the device-side plan follows the same rule (kvm_plan_tiles: hybrid = 2 with class C, 1 without).
    int hybrid_mode = 0;
    if (hybrid) {
        hybrid_mode = kvm_ed_dot_choose_classes(m, pl.n_tiles, pl.n_tiles_b, pl.n_tiles_c, ctx->num_cu, (int)ctx->opt_early_abandon);
        if (const int forced = (int)((ctx->opt_ed_force >> 12) & 3); forced == 1 || forced == 2) hybrid_mode = forced;   // KVM_OPT_ED_FORCE
        if (hybrid_mode == 1) {
            pl.n_tiles_b += pl.n_tiles_c;
            pl.n_tiles_c = 0;
        }
    }
    if (out.own) {  // capacity: every evaluated window may be an answer
        if ((rc = kvm_ensure(ctx, ctx->b_out_off, (size_t)pl.windows * 4 + 64))) return rc;
        if ((rc = kvm_ensure(ctx, ctx->b_out_dist, (size_t)pl.windows * 8 + 64))) return rc;
        out.d_off = (int32_t *)ctx->b_out_off.p;
        out.d_dist = (double *)ctx->b_out_dist.p;
        out.cap = pl.windows;
    }
    KVM_TP("  build_plan");

    // The interval list goes to the device first and the runs are built from it there (kvm_plan_runs_kernel, ~12 us)
    // while the host prepares the query block below.  Device layout: query block | left | right | runs | plan scratch.
    const size_t rbytes = (size_t)pl.n_runs * sizeof(kvm_run);
    const size_t plan_scratch = ((size_t)(k > 0 ? k : 0) / 256 + 1) * KVM_PLAN_SUMS_BYTES;  // block totals of kvm_launch_plan_runs
    const size_t rc_off = qbytes + lr_bytes + ((rbytes + 31) & ~(size_t)31) + ((plan_scratch + 31) & ~(size_t)31);
    const size_t hist_off = (rc_off + (chunks ? (size_t)pl.n_runs * sizeof(int2) : 0) + 15) & ~(size_t)15;  // cNSM: run_hist
    if ((rc = kvm_ensure(ctx, ctx->b_query, hist_off + (cnsm ? (size_t)pl.n_runs * sizeof(double) : 0) + 64))) return rc;
    kvm_run *d_runs_w = (kvm_run *)((char *)ctx->b_query.p + qbytes + lr_bytes);
    int2 *d_run_chunk = chunks ? (int2 *)((char *)ctx->b_query.p + rc_off) : nullptr;
    double *d_run_hist = cnsm ? (double *)((char *)ctx->b_query.p + hist_off) : nullptr;
    qd.run_hist = d_run_hist;
    if (rbytes) {
        int tile_log2 = 0;
        while ((1 << tile_log2) < tile_max) tile_log2++;
        // (the first plan kernel reads the interval list straight from the pinned staging buffer and leaves the device copy)
        HIPCHK(ctx, kvm_launch_plan_runs(pl.left, pl.right, (int32_t *)((char *)ctx->b_query.p + qbytes),
                                         (int32_t *)((char *)ctx->b_query.p + qbytes + lr_half), k, shift,
                                         ctx->n - m + 1, hybrid_mode, tile_log2, d_runs_w,
                                         (char *)ctx->b_query.p + qbytes + lr_bytes + ((rbytes + 31) & ~(size_t)31), stream,
                                         pl.chunk_lo, pl.chunk_hi, d_run_chunk));
    }

    KVM_TP("  plan launches");
    // correlation-form ED pruner: centred query and its chunk-granular prefix sums
    std::vector<double> qcen;
    std::vector<double2> qpref8;
    if (!dtw) {
        // (pruner-side quantities, guard-banded: the summation order is free, so four interleaved partial sums keep the
        // adders busy -- the plain sequential loops cost ~20 us of the host's critical path at m = 8192)
        double s4[4] = {0, 0, 0, 0};
        int i4 = 0;
        for (; i4 + 4 <= m; i4 += 4)
            for (int u = 0; u < 4; u++) s4[u] += q[i4 + u];
        for (; i4 < m; i4++) s4[0] += q[i4];
        qd.q_mean = ((s4[0] + s4[1]) + (s4[2] + s4[3])) / m;
        qcen.resize(m);
        qpref8.resize((size_t)nchunks8 + 1);
        static_assert(ED8_JC % 4 == 0, "chunk boundaries fall between groups of four");
        double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
        const double qm = qd.q_mean;
        int i = 0;
        for (; i < m; i += 4) {
            if (i % ED8_JC == 0) qpref8[(size_t)(i / ED8_JC)] = make_double2((a1[0] + a1[1]) + (a1[2] + a1[3]), (a2[0] + a2[1]) + (a2[2] + a2[3]));
            const int lim = std::min(4, m - i);
            for (int u = 0; u < lim; u++) {
                const double c = q[i + u] - qm;
                qcen[i + u] = c;
                a1[u] += c;
                a2[u] += c * c;
            }
        }
        const double p1 = (a1[0] + a1[1]) + (a1[2] + a1[3]), p2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
        qpref8[(size_t)nchunks8] = make_double2(p1, p2);
    }
    const size_t tbytes = (size_t)((retile || dtw) ? 3 * pl.n_tiles : pl.n_tiles + pl.n_tiles_b + pl.n_tiles_c) * sizeof(kvm_tile);
    // (KVM_SERIES_PAD spare doubles after q: the tiled ED kernel stages whole query chunks without bounds checks)
    if ((rc = kvm_ensure(ctx, ctx->b_tiles, tbytes + 64))) return rc;
    if ((rc = kvm_ensure(ctx, ctx->b_counters, KVM_CNT_TOTAL * sizeof(unsigned long long)))) return rc;
    // zero pads: after q (| qu | ql), after qc, and the tail of qT
    std::memset(hp + (size_t)m * 8 * (dtw ? 3 : 1), 0, KVM_SERIES_PAD * sizeof(double));
    if (!dtw) std::memset(hp + qmain + (size_t)m * 8, 0, KVM_SERIES_PAD * sizeof(double));
    std::memcpy(hp, q.data(), (size_t)m * 8);
    if (!dtw) {
        std::memcpy(hp + qmain, qcen.data(), (size_t)m * 8);
        std::memcpy(hp + qmain + qpref_off, qpref8.data(), qpref8.size() * 16);
        // the phases of the centred query (kvm_ed_dot.hip, the RING16 form of edd_body): qc[2t] | qc[2t + 1] | their sum, zero beyond m
        double *qe = (double *)(hp + qmain + qk_off), *qo = qe + nk, *qs = qo + nk;
        if (want_qk) std::memset(qe, 0, (size_t)3 * nk * 8);
        for (int t = 0; want_qk && 2 * t < m; t++) {
            const double e = qcen[(size_t)2 * t], o = 2 * t + 1 < m ? qcen[(size_t)2 * t + 1] : 0.0;
            qe[t] = e;
            qo[t] = o;
            qs[t] = e + o;
        }
    }
    const size_t qfirst = (cnsm && !dtw) ? qexact_off : qbytes;  // the exact-pass block is staged later (see below)
    if (dtw) {
        std::memcpy(hp + (size_t)m * 8, qu.data(), (size_t)m * 8);
        std::memcpy(hp + (size_t)m * 16, ql.data(), (size_t)m * 8);
        double *h_uo = (double *)(hp + qexact_off), *h_lo = h_uo + m, *h_qo = h_lo + m;
        int32_t *h_order = (int32_t *)(h_qo + m);
        for (int i = 0; i < m; i++) {
            const int o = ord[(size_t)i];
            h_order[i] = o;
            h_uo[i] = qu[(size_t)o];
            h_lo[i] = ql[(size_t)o];
            h_qo[i] = q[(size_t)o];
        }
    }
    KVM_TP("  query prep + staging");
    // scratch of the running-sum replay of the exact pass (cNSM): flags + {mean,std} per window, counts per run
    kvm_replay_ws rws;
    std::memset(&rws, 0, sizeof rws);
    if (cnsm && pl.windows > 0) {
        const size_t sbytes = (size_t)pl.windows * sizeof(double2);
        const size_t mbytes = (size_t)pl.n_runs * sizeof(int32_t);
        if ((rc = kvm_ensure(ctx, ctx->b_scratch, sbytes + 2 * mbytes + 64))) return rc;
        rws.runs = (const kvm_run *)((const char *)ctx->b_query.p + qbytes + lr_bytes);
        rws.n_runs = pl.n_runs;
        rws.stats = (double *)ctx->b_scratch.p;
        rws.stats_plane = pl.windows;
        rws.run_maxk = (int32_t *)((char *)ctx->b_scratch.p + sbytes);
        rws.run_list = rws.run_maxk + pl.n_runs;
        rws.run_list_n = (unsigned long long *)ctx->b_counters.p + KVM_CNT_REPLAY_RUNS;
    }
    // the query block (the tile list -- 16 B per tile -- is expanded from the runs on the GPU, by the kernel that also
    // zeroes the counters; for cNSM-ED the exact-pass block is filled in and copied further down, in stream order, before
    // its only reader -- the exact pass -- is launched)
    // (the upload rides on the expansion launch: one launch less per call -- a latency-bound call is a chain of launches)
    const kvm_run *d_runs = d_runs_w;
    if (rbytes) {
        if (hybrid)
            HIPCHK(ctx, kvm_launch_expand_tiles_hybrid(d_runs, pl.n_runs, hybrid_mode, (kvm_tile *)ctx->b_tiles.p,
                                                       (kvm_tile *)ctx->b_tiles.p + pl.n_tiles,
                                                       (kvm_tile *)ctx->b_tiles.p + pl.n_tiles + pl.n_tiles_b,
                                                       (unsigned long long *)ctx->b_counters.p, rws.run_maxk, stream,
                                                       ctx->b_query.p, hp, qfirst));
        else
            // (cNSM: also what the reference's running sums of each interval pass over -- guard of the pruners, kvm_hist_*)
            HIPCHK(ctx, kvm_launch_expand_tiles(d_runs, pl.n_runs, tile_max, (kvm_tile *)ctx->b_tiles.p,
                                                (unsigned long long *)ctx->b_counters.p, rws.run_maxk, stream,
                                                ctx->d_blockmax, d_run_hist, m, ctx->n, ctx->b_query.p, hp, qfirst));
    } else {
        HIPCHK(ctx, kvm_launch_upload(ctx->b_query.p, hp, qfirst, stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->b_counters.p, 0, KVM_CNT_TOTAL * sizeof(unsigned long long), stream));
    }
    qd.q = (const double *)ctx->b_query.p;
    qd.qc = dtw ? nullptr : (const double *)((const char *)ctx->b_query.p + qmain);
    qd.q_pref8 = dtw ? nullptr : (const double2 *)((const char *)ctx->b_query.p + qmain + qpref_off);
    qd.qk = want_qk ? (const double *)((const char *)ctx->b_query.p + qmain + qk_off) : nullptr;
    qd.qu = dtw ? qd.q + m : nullptr;
    qd.ql = dtw ? qd.q + 2 * (size_t)m : nullptr;
    unsigned long long *counters = (unsigned long long *)ctx->b_counters.p;
    const kvm_tile *d_tiles = (const kvm_tile *)ctx->b_tiles.p;
    const int64_t n_tiles = pl.n_tiles;

    if (kernel_ns) *kernel_ns = 0;
    ctx->last_kernel_ns = 0;
    ctx->timing_pending = false;
    KVM_TP("  H2D enqueue");
    if (n_tiles + pl.n_tiles_b + pl.n_tiles_c == 0) return KVM_OK;

    // (the default ED pruner on a host-planned list stamps its own begin and end: kvm_launch_ed_dot)
    const bool self_timed = ctx->opt_timing && !dtw && !retile && ed_kernel == 3;
    if (ctx->opt_timing && !self_timed) HIPCHK(ctx, hipEventRecord(ctx->ev0, stream));
    if (!dtw) {
        // pruning pass (fast kernels, guard band) -> survivors -> exact pass in the reference's operation order
        if ((rc = kvm_ensure(ctx, ctx->b_cands, (size_t)pl.windows * sizeof(kvm_surv) + 64))) return rc;
        kvm_surv *surv = (kvm_surv *)ctx->b_cands.p;
        if (retile) {
            kvm_tile *tA = (kvm_tile *)ctx->b_tiles.p + pl.n_tiles, *tB = tA + pl.n_tiles;
            if (retile_rsm)
                HIPCHK(ctx, kvm_launch_retile_rsm(ctx->d_series, qd, d_tiles, pl.n_tiles, tA, tB, counters + KVM_CNT_TILES_A,
                                                  stream));
            else
                HIPCHK(ctx, kvm_launch_retile_cnsm(ctx->d_series, ctx->d_prefix, qd, d_tiles, pl.n_tiles, tA, tB,
                                                   counters + KVM_CNT_TILES_A, stream));
            KVM_TP("  retile launch");
            // (scratch of the pruner's cross-workgroup hand-over on tiny lists: allocated once, its arrival counters zero between launches)
            if (!ctx->b_edx.p) {
                if ((rc = kvm_ensure(ctx, ctx->b_edx, kvm_ed_dot_x_bytes()))) return rc;
                HIPCHK(ctx, hipMemsetAsync((char *)ctx->b_edx.p + kvm_ed_dot_x_counters_at(), 0, kvm_ed_dot_x_bytes() - kvm_ed_dot_x_counters_at(), stream));
            }
            HIPCHK(ctx, kvm_launch_ed_dot(ctx->d_series, ctx->d_prefix, ctx->n, qd, tA, pl.n_tiles, tB, pl.n_tiles, tB, 0, surv,
                                          pl.windows, counters, ctx->num_cu, counters + KVM_CNT_TILES_A, out.d_off, out.d_dist, out.cap,
                                          stream, nullptr, nullptr, kvm_in_flight(ctx), ctx->opt_ed_force, &ctx->last_ed_shape,
                                          (double *)ctx->b_edx.p, (unsigned *)((char *)ctx->b_edx.p + kvm_ed_dot_x_counters_at())));
            KVM_TP("  ed_dot launch");
        }
        else if (ed_kernel == 3)
            HIPCHK(ctx, kvm_launch_ed_dot(ctx->d_series, ctx->d_prefix, ctx->n, qd, d_tiles, pl.n_tiles, d_tiles + pl.n_tiles,
                                          pl.n_tiles_b, d_tiles + pl.n_tiles + pl.n_tiles_b, pl.n_tiles_c, surv, pl.windows, counters, ctx->num_cu, nullptr, out.d_off, out.d_dist,
                                          out.cap, stream, self_timed ? ctx->ev0 : nullptr, self_timed ? ctx->ev1 : nullptr, kvm_in_flight(ctx), ctx->opt_ed_force, &ctx->last_ed_shape));
        else
            HIPCHK(ctx, kvm_launch_ed_wave(ctx->d_series, ctx->d_prefix, ctx->n, qd, d_tiles, n_tiles, surv,
                                           pl.windows, counters, stream));
        if (ctx->opt_timing && !self_timed) HIPCHK(ctx, hipEventRecord(ctx->ev1, stream));
        if (cnsm) {
            // cNSM-ED exact pass: the reference sums over the |zQ|-descending permutation (stable sort with
            // Double.compare(|o2|, |o1|), K/NormQueryEngine.java:443-452).  Computed here, while the pruning kernel runs.
            // Stable LSD radix sort on the bit patterns: for |x| the IEEE-754 pattern is monotone, NaN (canonicalised as
            // Double.compare does) sorts above +Inf, and complementing the key turns ascending into descending.
            double *zqs = (double *)(hp + qexact_off);
            int32_t *order = (int32_t *)(hp + qexact_off + (size_t)m * 8);
            {
                std::vector<uint64_t> ka((size_t)m), kb((size_t)m);
                std::vector<int32_t> ia((size_t)m), ib((size_t)m);
                for (int i = 0; i < m; i++) {
                    double a = std::fabs(q[i]);
                    uint64_t u;
                    std::memcpy(&u, &a, 8);
                    if (a != a) u = 0x7ff8000000000000ULL;
                    ka[(size_t)i] = ~u;
                    ia[(size_t)i] = i;
                }
                for (int pass = 0; pass < 8; pass++) {
                    const int sh = pass * 8;
                    size_t hist[257] = {0};
                    for (int i = 0; i < m; i++) hist[((ka[(size_t)i] >> sh) & 255u) + 1]++;
                    bool one_bucket = false;
                    for (int b = 1; b <= 256; b++) one_bucket |= hist[b] == (size_t)m;
                    if (one_bucket) continue;  // every key has the same digit: the pass would be the identity
                    for (int b = 0; b < 256; b++) hist[b + 1] += hist[b];
                    for (int i = 0; i < m; i++) {
                        const size_t d = hist[(ka[(size_t)i] >> sh) & 255u]++;
                        kb[d] = ka[(size_t)i];
                        ib[d] = ia[(size_t)i];
                    }
                    ka.swap(kb);
                    ia.swap(ib);
                }
                for (int i = 0; i < m; i++) {
                    order[i] = ia[(size_t)i];
                    zqs[i] = q[order[i]];
                }
            }
            HIPCHK(ctx, kvm_launch_upload((char *)ctx->b_query.p + qexact_off, hp + qexact_off, (size_t)m * 12, stream));
        }
        KVM_TP("  cnsm order + upload");
        const double *q_exact = cnsm ? (const double *)((const char *)ctx->b_query.p + qexact_off) : qd.q;
        const int32_t *d_order =
            cnsm ? (const int32_t *)((const char *)ctx->b_query.p + qexact_off + (size_t)m * 8) : nullptr;
        // KVM_OPT_UCR_LOOPS (executor mirrors only): UcrEdQueryExecutor's `sum < bsf` loop -- the windows whose ordered partial
        // sums may land exactly on eps^2 go through the literal loop behind the survivors (kvm_ucr.hip)
        kvm_surv *ties = nullptr;
        if (cnsm && ctx->opt_ucr_loops) {
            if ((rc = kvm_ensure(ctx, ctx->b_cands2, (size_t)pl.windows * sizeof(kvm_surv) + 64))) return rc;
            ties = (kvm_surv *)ctx->b_cands2.p;
            if (retile) {
                const kvm_tile *tA = (const kvm_tile *)ctx->b_tiles.p + pl.n_tiles, *tB = tA + pl.n_tiles;
                HIPCHK(ctx, kvm_launch_ucr_ed_screen(ctx->d_series, ctx->d_prefix, qd, q_exact, d_order, tA, pl.n_tiles, tB, pl.n_tiles,
                                                     counters + KVM_CNT_TILES_A, ties, pl.windows, counters, ctx->num_cu, stream));
            } else {
                HIPCHK(ctx, kvm_launch_ucr_ed_screen(ctx->d_series, ctx->d_prefix, qd, q_exact, d_order, d_tiles, n_tiles, nullptr, 0,
                                                     nullptr, ties, pl.windows, counters, ctx->num_cu, stream));
            }
        }
        HIPCHK(ctx, kvm_launch_exact_ed(ctx->d_series, qd, q_exact, d_order, surv, counters + KVM_CNT_CANDS1, pl.windows,
                                        rws, out.d_off, out.d_dist, out.cap, counters, ctx->num_cu, stream, ties,
                                        counters + KVM_CNT_UCR_TIES, ties ? pl.windows : 0));
        KVM_TP("  exact launch");
    } else {
        const int use_lb = (int)ctx->opt_lb_cascade;
        if ((rc = kvm_ensure(ctx, ctx->b_cands, (size_t)pl.windows * sizeof(kvm_cand) + 64))) return rc;
        if ((rc = kvm_ensure(ctx, ctx->b_cands2, (size_t)pl.windows * sizeof(kvm_cand) + 64))) return rc;  // ambiguous candidates
        if ((rc = ensure_envelope(ctx, rho, stream))) return rc;
        if ((rc = kvm_ensure(ctx, ctx->b_aux, (size_t)kvm_dtw_dp_max_waves(ctx->num_cu) * 8 * (size_t)rho * sizeof(double) + 64))) return rc;
        kvm_cand *c1 = (kvm_cand *)ctx->b_cands.p;
        kvm_dtw_aux &ax = dtw_ax;
        const double *d_dtw = (const double *)((const char *)ctx->b_query.p + qexact_off);
        ax.uo = d_dtw;
        ax.lo = d_dtw + m;
        ax.qo = d_dtw + 2 * (size_t)m;
        ax.order = (const int32_t *)(d_dtw + 3 * (size_t)m);
        ax.env_u = ctx->d_env_u;
        ax.env_l = ctx->d_env_l;
        ax.runs = d_runs_w;
        ax.n_runs = pl.n_runs;
        ax.run_chunk = d_run_chunk;
        ax.patch = (double *)ctx->b_aux.p;
        ax.ambig = (kvm_cand *)ctx->b_cands2.p;
        ax.cb_exact = nullptr;
        // a wave-parallel sum of m terms against the same terms added one by one (1e-12 + 5e-16 m), plus -- cNSM -- terms
        // formed with a reciprocal instead of a division per term (<= 6u m sqrt(sum), see dp_term_q in kvm_dtw.hip)
        ax.tau = 1e-12 + 5e-16 * (double)m + (cnsm ? 1e-15 * (double)m / std::sqrt(qd.eps2 > 0 ? qd.eps2 : 1.0) : 0.0);
        KVM_DBG_SYNC(ctx, stream, "plan / upload / envelope");
        if (use_lb) {
            // (alpha, beta) filter + LB_Kim for every candidate, survivors re-tiled on the device; then LB_Keogh on the
            // query envelope, in the reference's order, as a screen
            kvm_tile *t2 = (kvm_tile *)ctx->b_tiles.p + n_tiles;
            HIPCHK(ctx, kvm_launch_retile_dtw(ctx->d_series, ctx->d_prefix, qd, d_tiles, n_tiles, t2, counters, stream));
            KVM_DBG_SYNC(ctx, stream, "retile_dtw");
            HIPCHK(ctx, kvm_launch_dtw_screen(ctx->d_series, ctx->d_prefix, qd, ax, t2, n_tiles, counters + KVM_CNT_TILES_A, c1,
                                              counters, stream));
            KVM_DBG_SYNC(ctx, stream, "dtw_screen");
        } else {
            HIPCHK(ctx, kvm_launch_dtw_filter(ctx->d_series, ctx->d_prefix, ctx->n, qd, d_tiles, n_tiles, 0, c1, counters,
                                              stream));
        }
        const unsigned long long *dp_n = counters + KVM_CNT_CANDS1;
        // cNSM: replay the reference's running sums for the candidates that reach the deciding stage (exact mean/std and
        // exact (alpha, beta) decision), so that every value it forms is the reference's own
        if (cnsm)
            HIPCHK(ctx, kvm_launch_exact_stats(ctx->d_series, qd, c1, dp_n, pl.windows, rws, counters, ctx->num_cu, stream));
        KVM_DBG_SYNC(ctx, stream, "exact_stats");
        if ((rc = kvm_ensure(ctx, ctx->b_plans, kvm_dtw_plan_bytes(pl.windows) + 64))) return rc;
        HIPCHK(ctx, kvm_launch_dtw_dp(ctx->d_series, ctx->n, qd, ax, c1, dp_n, pl.windows, ctx->b_plans.p, out.d_off, out.d_dist,
                                      out.cap, counters, ctx->num_cu, (int)ctx->opt_dp_kernel, stream,
                                      ctx->opt_timing ? ctx->ev0 : nullptr));  // (timing: the DP kernel alone)
        KVM_DBG_SYNC(ctx, stream, "dtw_dp");
        dtw_qd = qd;
        if (cnsm && ctx->opt_ucr_loops) {
            out.rws = rws;
            out.tiles = d_tiles;
            out.n_tiles = n_tiles;
            out.windows = pl.windows;
            out.ucr_tables_ready = false;
        }
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_staged, stream));  // every copy out of h_pinned has been enqueued before this point
    ctx->staged_pending = true;
    if (ctx->opt_timing) {
        if (dtw) HIPCHK(ctx, hipEventRecord(ctx->ev1, stream));  // (ED: ev1 was recorded right after the ED kernel)
        ctx->timing_pending = true;  // read by kvm_finish_timing() after the call's synchronisation / in kvm_last_kernel_ns
    }
    return KVM_OK;
}

// DTW: the candidates the DP kernel could not decide within rounding (counters[KVM_CNT_AMBIG] of them, in ctx->b_cands2)
// go through the sequential reference-ordered path (kvm_dtw_ref.hip), in batches that fit a bounded slot buffer (2m
// doubles per candidate), each followed by a DP run that reads the exact cb[] arrays.  Answers are appended through the
// same counters.  Called after the main pipeline's counters have reached the host (one extra synchronisation, only when
// such candidates exist: never on continuous data, routinely on integer-valued data and for eps = 0).
static int dtw_resolve_ambiguous(kvm_ctx *ctx, const kvm_query_dev &qd, const kvm_dtw_aux &ax, int64_t n_ambig, verify_out &out,
                                 hipStream_t stream) {
    if (n_ambig <= 0) return KVM_OK;
    const size_t per = (size_t)2 * qd.m * sizeof(double);
    // (128 MiB of cb slots per batch: 2 048 candidates at m = 4096 -- a round of the chains' wave slots; the buffer is grow-only
    //  and stays with the context, so a tie-heavy query must not leave a gigabyte behind)
    int64_t batch = std::max<int64_t>(1, (int64_t)(((size_t)128 << 20) / per));
    if (batch > n_ambig) batch = n_ambig;
    int rc = kvm_ensure(ctx, ctx->b_aux2, (size_t)batch * per + 64);
    if (rc) return rc;
    unsigned long long *counters = (unsigned long long *)ctx->b_counters.p;
    kvm_cand *cands = (kvm_cand *)ctx->b_cands2.p;
    double *slots = (double *)ctx->b_aux2.p;
    kvm_dtw_aux ax2 = ax;
    ax2.cb_exact = slots;
    ax2.ambig = nullptr;
    if (kvm_trace_on())
        fprintf(stderr, "[kvm] dtw sequential path: %lld candidates, m %d rho %d, batches of %lld\n", (long long)n_ambig, qd.m,
                qd.rho, (long long)batch);
    // KVM_OPT_UCR_LOOPS (UcrDtwQueryExecutor's never-cleared cb1 / cb2, kvm_ucr.hip): exit depths of every window of the scan
    // once per query, then per candidate the terms earlier windows left behind
    const bool stale = out.tiles != nullptr;
    int32_t *depth1 = nullptr, *depth2 = nullptr, *tmax1 = nullptr, *tmax2 = nullptr;
    if (stale) {
        if ((rc = kvm_ensure(ctx, ctx->b_ucr, ((size_t)2 * out.windows + 2 * out.n_tiles) * sizeof(int32_t) + 64))) return rc;
        depth1 = (int32_t *)ctx->b_ucr.p;
        depth2 = depth1 + out.windows;
        tmax1 = depth2 + out.windows;
        tmax2 = tmax1 + out.n_tiles;
    }
    for (int64_t off = 0; off < n_ambig; off += batch) {
        const int64_t nb = std::min(batch, n_ambig - off);
        if (stale) {
            HIPCHK(ctx, kvm_launch_ucr_dtw_stale(ctx->d_series, qd, ax, out.tiles, out.n_tiles, out.rws, depth1, depth2, tmax1, tmax2,
                                                 cands + off, nb, slots, out.ucr_tables_ready, stream));
            out.ucr_tables_ready = true;
            KVM_DBG_SYNC(ctx, stream, "ucr_dtw_stale");
        }
        HIPCHK(ctx, kvm_launch_dtw_ref_chains(ctx->d_series, qd, ax, cands + off, nb, slots, stream, stale));
        KVM_DBG_SYNC(ctx, stream, "dtw_ref_chains");
        HIPCHK(ctx, kvm_launch_set_u64(counters + KVM_CNT_CANDS2, (unsigned long long)nb, counters + KVM_CNT_QUEUE, 0ULL, stream));
        HIPCHK(ctx, kvm_launch_dtw_dp(ctx->d_series, ctx->n, qd, ax2, cands + off, counters + KVM_CNT_CANDS2, nb, nullptr,
                                      out.d_off, out.d_dist, out.cap, counters, ctx->num_cu, (int)ctx->opt_dp_kernel,
                                      stream));
        KVM_DBG_SYNC(ctx, stream, "dtw_dp (exact cb)");
    }
    return KVM_OK;
}

// answers.sort(Comparator.comparing(Pair::getSecond)) -- stable on an offset-ascending list
// (K/QueryEngine.java:373) == sort by (distance, offset).
// Packed records; above a few thousand answers a stable LSD radix sort (offset digits first, then the bit pattern
// of the non-negative distance, 11 bits per pass, passes whose digit is the same for every record skipped) instead
// of a comparison sort: 0.46 -> ~0.1 ms for 10^4 answers.
static void sort_answers_into(const int32_t *off, const double *dist, int64_t cnt, kvm_result &r) {
    struct ans_rec {
        double d;
        int32_t o;
    };
    std::vector<ans_rec> recs((size_t)cnt);
    for (int64_t i = 0; i < cnt; i++) recs[(size_t)i] = ans_rec{dist[(size_t)i], off[(size_t)i]};
    if (cnt < 2048) {
        std::sort(recs.begin(), recs.end(), [](const ans_rec &a, const ans_rec &b) { return a.d != b.d ? a.d < b.d : a.o < b.o; });
    } else {
        std::vector<ans_rec> tmp((size_t)cnt);
        ans_rec *src = recs.data(), *dst = tmp.data();
        auto key_of = [](const ans_rec &r, int pass) -> uint32_t {
            if (pass < 3) return ((uint32_t)r.o >> (11 * pass)) & 2047u;  // offsets are positive int32
            uint64_t u;
            std::memcpy(&u, &r.d, 8);  // distances are >= 0 and never NaN: the bit pattern is monotone
            return (uint32_t)(u >> (11 * (pass - 3))) & 2047u;
        };
        for (int pass = 0; pass < 9; pass++) {
            size_t hist[2049] = {0};
            for (int64_t i = 0; i < cnt; i++) hist[key_of(src[i], pass) + 1]++;
            bool uniform = false;
            for (int b = 1; b <= 2048; b++) uniform |= hist[b] == (size_t)cnt;
            if (uniform) continue;
            for (int b = 0; b < 2048; b++) hist[b + 1] += hist[b];
            for (int64_t i = 0; i < cnt; i++) dst[hist[key_of(src[i], pass)]++] = src[i];
            std::swap(src, dst);
        }
        if (src != recs.data()) std::memcpy(recs.data(), src, (size_t)cnt * sizeof(ans_rec));
    }
    r.off.resize((size_t)cnt);
    r.dist.resize((size_t)cnt);
    for (int64_t i = 0; i < cnt; i++) {
        r.off[(size_t)i] = recs[(size_t)i].o;
        r.dist[(size_t)i] = recs[(size_t)i].d;
    }
}
// The group layer's result (kvm_sharded.cpp): the merged answers of all shards, in the order a single call returns them.
kvm_result *kvm_result_from_pairs(const int32_t *off, const double *dist, int64_t cnt, int64_t candidates, double elapsed_ms) {
    try {
        std::unique_ptr<kvm_result> r(new kvm_result());
        sort_answers_into(off, dist, cnt, *r);
        r->candidates = candidates;
        r->elapsed_ms = elapsed_ms;
        return r.release();
    } catch (...) {
        return nullptr;
    }
}

static int verify_host(kvm_ctx *ctx, int variant, const double *query, int32_t m, const int32_t *left,
                       const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho, double alpha,
                       double beta, kvm_result **result) {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!result) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "null result pointer");
    *result = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto tp_ = t0;
    plan pl;
    int rc;
    verify_out out{nullptr, nullptr, 0, true};
    int64_t kernel_ns = 0;
    kvm_query_dev dtw_qd;
    kvm_dtw_aux dtw_ax;
    kvm_enqueue_lock enq(ctx);   // (KVM_OPT_SERIALIZE_ENQUEUE)
    rc = run_variant(ctx, variant, query, m, left, right, k, shift, epsilon, rho, alpha, beta, pl, out, ctx->stream,
                     &kernel_ns, dtw_qd, dtw_ax);
    if (rc) return rc;
    const int64_t worst = out.cap;
    const bool is_dtw = variant == V_RSM_DTW || variant == V_CNSM_DTW;
    int64_t n_sequential = 0;
    KVM_TP("run_variant (launch)");
    // One synchronisation in the common case: the counters and the first KVM_SPEC_ANSWERS answers are copied back
    // together (answers are rare); a second copy only when there are more.
    static_assert(KVM_CNT_TOTAL <= 256 && (KVM_SPEC_ANSWERS % 2) == 0, "readback kernel layout");
    unsigned long long hc[KVM_CNT_TOTAL] = {0};
    const int64_t spec = std::min<int64_t>(KVM_SPEC_ANSWERS, worst);
    std::vector<int32_t> off;
    std::vector<double> dist;
    for (int pass = 0; pass < 2 && pl.n_tiles + pl.n_tiles_b + pl.n_tiles_c > 0; pass++) {
        if (!ctx->h_readback)
            HIPCHK(ctx, hipHostMalloc(&ctx->h_readback, KVM_CNT_TOTAL * 8 + KVM_SPEC_ANSWERS * 12 + 64, hipHostMallocDefault));
        enq.again();
        HIPCHK(ctx, kvm_launch_readback((const unsigned long long *)ctx->b_counters.p, out.d_off, out.d_dist, spec,
                                        ctx->h_readback, ctx->stream));
        enq.release();
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(hc, ctx->h_readback, sizeof hc);
        if (pass == 0 && is_dtw && hc[KVM_CNT_AMBIG] > 0) {
            // candidates within rounding of a tie with eps^2: sequential reference-ordered path, then read back again
            n_sequential = (int64_t)hc[KVM_CNT_AMBIG];
            enq.again();
            if ((rc = dtw_resolve_ambiguous(ctx, dtw_qd, dtw_ax, n_sequential, out, ctx->stream))) return rc;
            continue;
        }
        const int64_t got = std::min<int64_t>((int64_t)hc[KVM_CNT_ANSWERS], spec);
        const int32_t *hoff = (const int32_t *)((const unsigned long long *)ctx->h_readback + KVM_CNT_TOTAL);
        const double *hdist = (const double *)(hoff + ((spec + 1) & ~(int64_t)1));
        off.assign(hoff, hoff + got);
        dist.assign(hdist, hdist + got);
        break;
    }
    KVM_TP("sync + counters D2H");
    if (kvm_trace_on())
        fprintf(stderr, "[kvm] tiles %lld+%lld+%lld  retiled A %llu B %llu C %llu  pruner survivors %llu\n", (long long)pl.n_tiles,
                (long long)pl.n_tiles_b, (long long)pl.n_tiles_c, hc[KVM_CNT_TILES_A], hc[KVM_CNT_TILES_B], hc[KVM_CNT_TILES_C],
                hc[KVM_CNT_CANDS1]);
    const int64_t cnt = (int64_t)hc[KVM_CNT_ANSWERS];
    if (cnt > worst) return kvm_fail(ctx, KVM_ERR_DEVICE, "answer buffer overflow (internal error)");
    std::unique_ptr<kvm_result> r_owner(new kvm_result());  // (released into *result at the end; freed on every error return)
    kvm_result *r = r_owner.get();
    off.resize((size_t)cnt);
    dist.resize((size_t)cnt);
    if (cnt > spec) {
        // more answers than the readback kernel carries: the rest through a pinned buffer (grow-only)
        const size_t rest = (size_t)(cnt - spec);
        if (ctx->h_readback2_cap < rest * 12 + 64) {
            if (ctx->h_readback2) HIPCHK(ctx, hipHostFree(ctx->h_readback2));
            ctx->h_readback2 = nullptr;
            ctx->h_readback2_cap = 0;
            const size_t want = std::max<size_t>(rest * 12 + 64, (size_t)1 << 20);
            HIPCHK(ctx, hipHostMalloc(&ctx->h_readback2, want, hipHostMallocDefault));
            ctx->h_readback2_cap = want;
        }
        double *pd = (double *)ctx->h_readback2;
        int32_t *po = (int32_t *)(pd + rest);
        enq.again();
        HIPCHK(ctx, hipMemcpyAsync(pd, out.d_dist + spec, rest * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(po, out.d_off + spec, rest * 4, hipMemcpyDeviceToHost, ctx->stream));
        enq.release();
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(dist.data() + spec, pd, rest * 8);
        std::memcpy(off.data() + spec, po, rest * 4);
    }
    KVM_TP("answers D2H");
    sort_answers_into(off.data(), dist.data(), cnt, *r);
    KVM_TP("sort answers");
    r->candidates = pl.cnt_candidate;
    r->stats[KVM_STAT_WINDOWS] = pl.windows;
    const bool cnsm = variant == V_CNSM_ED || variant == V_CNSM_DTW;
    const bool dtw = variant == V_RSM_DTW || variant == V_CNSM_DTW;
    r->stats[KVM_STAT_PASS_FILTER] = cnsm ? (int64_t)hc[KVM_CNT_PASS_FILTER] : pl.windows;
    r->stats[KVM_STAT_RECHECKED] = (int64_t)hc[KVM_CNT_RECHECKED];
    r->stats[KVM_STAT_DTW_SEQUENTIAL] = n_sequential;
    if (dtw) {
        r->stats[KVM_STAT_PASS_LB_KIM] = (int64_t)hc[KVM_CNT_PASS_KIM];
        r->stats[KVM_STAT_PASS_LB_KEOGH] = ctx->opt_lb_cascade ? (int64_t)hc[KVM_CNT_PASS_KEOGH] : (int64_t)hc[KVM_CNT_PASS_KIM];
        r->stats[KVM_STAT_PASS_LB_KEOGH2] = (int64_t)hc[KVM_CNT_PASS_KEOGH2];  // DP runs started
    }
    kvm_finish_timing(ctx);
    kernel_ns = ctx->last_kernel_ns;
    r->stats[KVM_STAT_KERNEL_NS] = kernel_ns;
    auto t1 = std::chrono::steady_clock::now();
    r->elapsed_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    *result = r_owner.release();
    return KVM_OK;
}

extern "C" int kvm_verify_rsm_ed(kvm_ctx *ctx, const double *query, int32_t m, const int32_t *left,
                                 const int32_t *right, int32_t k, int32_t shift, double epsilon,
                                 kvm_result **result) {
    return kvm_guard<int>(ctx, [&]() -> int {
    return verify_host(ctx, V_RSM_ED, query, m, left, right, k, shift, epsilon, 0, 1.0, 0.0, result);
    });
}
extern "C" int kvm_verify_cnsm_ed(kvm_ctx *ctx, const double *query, int32_t m, const int32_t *left,
                                  const int32_t *right, int32_t k, int32_t shift, double epsilon, double alpha,
                                  double beta, kvm_result **result) {
    return kvm_guard<int>(ctx, [&]() -> int {
    return verify_host(ctx, V_CNSM_ED, query, m, left, right, k, shift, epsilon, 0, alpha, beta, result);
    });
}
extern "C" int kvm_verify_rsm_dtw(kvm_ctx *ctx, const double *query, int32_t m, const int32_t *left,
                                  const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho,
                                  kvm_result **result) {
    return kvm_guard<int>(ctx, [&]() -> int {
    return verify_host(ctx, V_RSM_DTW, query, m, left, right, k, shift, epsilon, rho, 1.0, 0.0, result);
    });
}
extern "C" int kvm_verify_cnsm_dtw(kvm_ctx *ctx, const double *query, int32_t m, const int32_t *left,
                                   const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho,
                                   double alpha, double beta, kvm_result **result) {
    return kvm_guard<int>(ctx, [&]() -> int {
    return verify_host(ctx, V_CNSM_DTW, query, m, left, right, k, shift, epsilon, rho, alpha, beta, result);
    });
}

extern "C" int kvm_verify_device(kvm_ctx *ctx, int32_t variant, const double *query, int32_t m, const int32_t *left,
                                 const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho,
                                 double alpha, double beta, int32_t *d_out_off, double *d_out_dist, int64_t cap,
                                 int64_t *d_out_count, int64_t *cnt_candidate, void *stream) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (variant < 0 || variant > 3 || !d_out_off || !d_out_dist || !d_out_count || cap < 0)
        return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "bad kvm_verify_device arguments");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // Everything this call enqueues -- the pipeline, the sequential tie path's kernels, the copy of the answer count out of
    // the counters -- reads the context's device scratch: the next call, possibly on another stream, must be ordered behind
    // all of it before it reuses the buffers.  That is a device-side wait on ev_done (run_variant, ensure_envelope), so the
    // event is recorded on EVERY way out of this call, a failure half-way down the pipeline included (its kernels are still
    // running); the HOST only ever waits for ev_staged, recorded behind the last read of the pinned staging buffer, so it
    // can prepare the next call while this one still runs.
    struct done_guard {
        kvm_ctx *c;
        hipStream_t s;
        ~done_guard() {
            // (armed by what run_variant left behind: it clears done_pending once this call's stream is ordered behind the previous
            // call, before it enqueues anything -- so a cleared flag means "this call may have enqueued", whether it then
            // returned, failed half-way or threw; a flag still set means an argument error that enqueued nothing, and the
            // previous call's mark stays in place)
            if (!c->done_pending && hipSetDevice(c->device) == hipSuccess && hipEventRecord(c->ev_done, s) == hipSuccess) {
                c->done_stream = s;
                c->done_pending = true;
            }
        }
    };
    plan pl;
    verify_out out{d_out_off, d_out_dist, cap, false};
    kvm_query_dev dtw_qd;
    kvm_dtw_aux dtw_ax;
    kvm_enqueue_lock enq(ctx);   // (KVM_OPT_SERIALIZE_ENQUEUE: see the top of this file; declared before the guard
    // below, so that the guard's event is recorded while the lock is still held)
    done_guard guard{ctx, s};    // (before run_variant: an exception out of it -- std::bad_alloc -- leaves through the guard too)
    int rc = run_variant(ctx, variant, query, m, left, right, k, shift, epsilon, rho, alpha, beta, pl, out, s, nullptr, dtw_qd,
                         dtw_ax);
    if (rc) return rc;
    if (cnt_candidate) *cnt_candidate = pl.cnt_candidate;
    if (variant >= V_RSM_DTW && pl.n_tiles + pl.n_tiles_b + pl.n_tiles_c > 0) {
        // DTW: the number of candidates that need the sequential reference-ordered path is only known on the device;
        // this one small read is the only point where a kvm_verify_device call waits for the stream
        unsigned long long n_ambig = 0;
        HIPCHK(ctx, hipMemcpyAsync(&n_ambig, (unsigned long long *)ctx->b_counters.p + KVM_CNT_AMBIG, sizeof n_ambig,
                                   hipMemcpyDeviceToHost, s));
        enq.release();
        HIPCHK(ctx, hipStreamSynchronize(s));
        enq.again();
        if (n_ambig > 0 && (rc = dtw_resolve_ambiguous(ctx, dtw_qd, dtw_ax, (int64_t)n_ambig, out, s))) return rc;
    }
    // (the answer count, by a one-thread kernel: a copy command costs the enqueueing thread ~10 us under a lock all streams share)
    HIPCHK(ctx, kvm_launch_copy_u64((unsigned long long *)d_out_count,
                                    pl.n_tiles + pl.n_tiles_b + pl.n_tiles_c == 0 ? nullptr
                                                                                  : (unsigned long long *)ctx->b_counters.p + KVM_CNT_ANSWERS, s));
    return KVM_OK;   // (~done_guard records ev_done behind everything enqueued above)
    });
}

extern "C" int kvm_pack_answers_device(kvm_ctx *ctx, const int32_t *d_off, const double *d_dist, const int64_t *d_count,
                                       int64_t pack_cap, double *d_packed, void *stream) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!d_off || !d_dist || !d_count || !d_packed || pack_cap < 0)
        return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "bad kvm_pack_answers_device arguments");
    kvm_enqueue_lock enq(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, kvm_launch_pack_answers(d_off, d_dist, d_count, pack_cap, (double2 *)d_packed,
                                        stream ? (hipStream_t)stream : ctx->stream));
    return KVM_OK;
    });
}

// the failure mark of the multi-GPU exchange: {-1, 0} in front of a slot's packed block (kvm_run_device_jobs)
int kvm_poison_packed_device(kvm_ctx *ctx, double *d_packed, void *stream) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx || !d_packed) return KVM_ERR_INVALID_HANDLE;
    static const double mark[2] = {-1.0, 0.0};
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(d_packed, mark, sizeof mark, hipMemcpyHostToDevice, stream ? (hipStream_t)stream : ctx->stream));
    return KVM_OK;
    });
}

// kvm_run_device_jobs: a slot's `ready` event, behind the job's kernels on the slot's stream
int kvm_record_event_device(kvm_ctx *ctx, void *event, void *stream) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx || !event) return KVM_ERR_INVALID_HANDLE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventRecord((hipEvent_t)event, stream ? (hipStream_t)stream : ctx->stream));
    return KVM_OK;
    });
}

// ---- results ----------------------------------------------------------------------------------------------
extern "C" int64_t kvm_result_count(const kvm_result *r) { return r ? (int64_t)r->off.size() : 0; }
extern "C" int64_t kvm_result_candidates(const kvm_result *r) { return r ? r->candidates : 0; }
extern "C" double kvm_result_elapsed_ms(const kvm_result *r) { return r ? r->elapsed_ms : 0.0; }
extern "C" const int32_t *kvm_result_offsets(const kvm_result *r) { return r ? r->off.data() : nullptr; }
extern "C" const double *kvm_result_distances(const kvm_result *r) { return r ? r->dist.data() : nullptr; }
extern "C" int64_t kvm_result_stat(const kvm_result *r, int32_t which) {
    return (r && which >= 0 && which < KVM_STAT_COUNT) ? r->stats[which] : 0;
}
extern "C" void kvm_result_free(kvm_result *r) { delete r; }

