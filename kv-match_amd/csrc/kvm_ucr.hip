// kvm_ucr.hip -- the loop semantics of the reference's index-free executors where they differ from the engines'
// (KVM_OPT_UCR_LOOPS; U/ = src/main/java/cn/edu/fudan/dsm/kvmatch/experiments/ucr/).  Only the executor mirrors
// (kvmatch/ucr.py, host/kvmatch_engines.hpp) set the option; the four engine entry points never come here.
//
// 1. UcrEdQueryExecutor.distance() (U/UcrEdQueryExecutor.java:91-98) runs `for (i = 0; i < m && sum < bsf; i++)` where the
//    engines run `while (j < m && dist <= bsf)` (K/NormQueryEngine.java:513), and run() accepts `dist <= Epsilon * Epsilon`
//    (:165).  So a partial sum -- in the executor's |zQ|-descending order -- that lands EXACTLY on eps^2 before the last
//    term ends the loop, and the window is reported at distance sqrt(eps^2) whatever its real distance; with eps^2 == 0
//    the loop never starts and every window that passes the (Alpha, Beta) test is reported at distance 0.  Everything else
//    is the engine's answer (a clean crossing is rejected by both; a window whose loop runs to the end gets the same sum).
//      kvm_ucr_ed_screen_kernel : one lane per window of the filter-passing tiles, the terms in the executor's order with the
//        pruning arithmetic (prefix-sum mean / sigma): the first partial >= eps^2 - slack; above eps^2 + slack -> a clean
//        crossing, nothing to add; inside the band (or undecided after UCR_SCREEN_TERMS terms, or sigma degenerate) -> a tie
//        candidate.  On continuous data the band is empty.
//      kvm_ucr_ed_tie_kernel    : one wavefront per tie candidate, the executor's loop literally -- replayed mean / std
//        (the executor's single chain of running sums = the cNSM replay of its run), true division, terms added one by one
//        by lane 0 -- and, where it stops on `sum == bsf`, the ENGINE's sum is carried on over the remaining terms: if it
//        ends above eps^2 the engine does not report the window, so it is appended here at sqrt(bsf); if it stays on eps^2
//        (terms too small to move the sum) the engine reports the same window at the same distance itself.
//
// 2. UcrDtwQueryExecutor allocates cb1 / cb2 once (U/UcrDtwQueryExecutor.java:157-161) and never clears them, while the
//    engines allocate fresh arrays per window (K/NormQueryEngineDtw.java:559,571).  lbKeoghCumulative /
//    lbKeoghDataCumulative write cb[order[i]] for the i they reach; when a loop stops early with its bound EXACTLY on
//    eps^2 the window goes on and the DP's cb[] is built from an array whose tail still holds what EARLIER windows wrote:
//    for position order[i], i >= this window's exit depth, the term of the last earlier window (in scan order, across the
//    EPOCH buffers) whose loop got past depth i -- zero if none did.  Only the candidates of the sequential tie path
//    (kvm_dtw_ref.hip) can see such entries.  For them:
//      kvm_ucr_dtw_depth_kernel : for EVERY window of the scan, in the reference's arithmetic (replayed mean / std): did it
//        reach lbKeoghCumulative / lbKeoghDataCumulative, and how many iterations did each loop make (0 .. m);
//      kvm_ucr_dtw_stale_kernel : one wavefront per tie candidate walks back over the earlier windows (block maxima of
//        the depths let it skip) and fills the candidate's cb1 / cb2 slots with the terms those windows left behind;
//      kvm_dtw_ref_chains_kernel then runs with `keep_slots`: it does not zero the arrays first.
#include "kvm_internal.h"
#include "kvm_dtw_ref.h"

#define UCR_SCREEN_TERMS 512

// ---- 1. UcrEdQueryExecutor ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kvm_ucr_ed_screen_kernel(const double *__restrict__ series, const double2 *__restrict__ prefix,
                                                                kvm_query_dev qd, const double *__restrict__ zqs,
                                                                const int32_t *__restrict__ order,
                                                                const kvm_tile *__restrict__ tilesA, int64_t nA,
                                                                const kvm_tile *__restrict__ tilesB, int64_t nB,
                                                                const unsigned long long *__restrict__ d_counts,
                                                                kvm_surv *__restrict__ ties, int64_t cap,
                                                                unsigned long long *__restrict__ counters) {
    if (d_counts) {
        nA = (int64_t)d_counts[0];
        nB = (int64_t)d_counts[1];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = qd.m;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < nA + nB; t += (int64_t)gridDim.x * 4) {
        const kvm_tile tile = t < nA ? tilesA[t] : tilesB[t - nA];
        const bool have = lane < tile.count;
        const int64_t w0 = (int64_t)tile.start - 1 + (have ? lane : 0);
        double mean = 0.0, sd = 1.0;
        bool alive = false, tie = false;
        double lim = 0.0, inv = 0.0, nmi = 0.0;
        if (have) {
            const kvm_stat_guard sg = kvm_window_mean_std_g(prefix, series, w0, m, mean, sd, kvm_run_hist_of(qd, tile.run));
            if (kvm_cnsm_filter_guarded(mean, sd, qd.meanQ, qd.stdQ, qd.alpha, qd.beta, sg)) {
                if (!(sd == sd) || !(sd > 0.0) || kvm_guard_degenerate(sg.g)) {
                    tie = true;  // nothing can be said with this sigma: the literal loop decides
                } else {
                    lim = qd.eps2 + kvm_eps2_slack(qd.eps2, m, sg.g, 16.0);
                    inv = 1.0 / sd;
                    nmi = -mean * inv;
                    alive = true;
                }
            }
        }
        const double lo_thr = 2.0 * qd.eps2 - lim;  // eps^2 - slack
        if (alive && !(lo_thr > 0.0)) {             // (the executor tests `sum < bsf` before the first term)
            alive = false;
            tie = true;
        }
        const double *__restrict__ xw = series + w0;
        double p = 0.0;
        const int m_scr = min(m, UCR_SCREEN_TERMS);
        for (int i0 = 0; i0 < m_scr && __any(alive); i0 += 8) {
            double xv[8], zq[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i_ = min(i0 + u, m - 1);
                xv[u] = xw[order[i_]];
                zq[u] = zqs[i_];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const double z = __builtin_fma(xv[u], inv, nmi), d = z - zq[u];
                const double pn = __builtin_fma(d, d, p);
                const bool step = alive && i0 + u < m;
                p = step ? pn : p;
                if (step && !(p < lo_thr)) {  // (NaN: stops, and counts as inside the band)
                    tie = !(p > lim);
                    alive = false;
                }
            }
        }
        // a window that never reached the band: over the whole window -> the executor's loop runs to the end like the engine's
        // (nothing to add); after UCR_SCREEN_TERMS terms of a longer window -> undecided, the literal loop looks at it
        if (alive && m_scr < m) tie = true;
        const unsigned long long bal = __ballot(tie);
        if (bal == 0ULL) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&counters[KVM_CNT_UCR_TIES], (unsigned long long)__popcll(bal));
        base = kvm_uniform_u64(base);
        if (tie) {
            const unsigned long long idx = base + __popcll(bal & ((1ULL << lane) - 1ULL));
            if ((int64_t)idx < cap) ties[idx] = kvm_surv{tile.start + lane, tile.run};
        }
    }
}

__global__ __launch_bounds__(256) void kvm_ucr_ed_tie_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                             const double *__restrict__ zqs, const int32_t *__restrict__ order,
                                                             const kvm_surv *__restrict__ ties,
                                                             const unsigned long long *__restrict__ d_n, int64_t cap_ties,
                                                             kvm_replay_ws ws, int32_t *__restrict__ out_off,
                                                             double *__restrict__ out_dist, int64_t cap,
                                                             unsigned long long *__restrict__ counters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = qd.m;
    const double bsf = qd.eps2;
    unsigned long long n = *d_n;
    if ((int64_t)n > cap_ties) n = (unsigned long long)cap_ties;
    for (unsigned long long idx = (unsigned long long)blockIdx.x * 4 + wave; idx < n; idx += (unsigned long long)gridDim.x * 4) {
        const kvm_surv sv = ties[idx];
        const kvm_run rn = ws.runs[sv.run];
        const int64_t wi = rn.win_base + (sv.start - rn.start);
        const double mean = ws.stats[wi] / m;                                          // U/UcrEdQueryExecutor.java:152
        const double std = sqrt(ws.stats[ws.stats_plane + wi] / m - mean * mean);      // :153-154
        if (!kvm_cnsm_filter(mean, std, qd.meanQ, qd.stdQ, qd.alpha, qd.beta)) continue;  // :156 (wave-uniform)
        const double *__restrict__ x = series + ((int64_t)sv.start - 1);
        // state of the literal loop: 0 = adding (`sum < bsf` held so far), 1 = the executor has stopped on sum == bsf and the
        // engine's loop is followed to its end, 2 = decided
        int state = 0;
        bool report = false;
        double sum = 0.0;
        if (!(sum < bsf)) state = (sum == bsf) ? 1 : 2;                                // :93, before the first term
        for (int k0 = 0; k0 < m && state != 2; k0 += 64) {
            const int k = min(k0 + lane, m - 1);
            const double xz = (x[order[k]] - mean) / std;                              // :94
            const double term = (xz - zqs[k]) * (xz - zqs[k]);                        // :95
            const int cnt = min(64, m - k0);
            for (int l = 0; l < cnt && state != 2; l++) {                              // (uniform: every lane follows lane 0's state)
                const double tl = kvm_readlane_f64(term, l);
                if (state == 0) {
                    sum += tl;                                                         // :95, in order
                    if (!(sum < bsf)) state = (sum == bsf && k0 + l + 1 < m) ? 1 : 2;  // stopped before the last term on a tie?
                } else {
                    // the engine's loop (`dist <= bsf`, K/NormQueryEngine.java:513) goes on from the same sum: it reports the
                    // window itself iff the sum is still <= bsf at the end (then: same offset, same distance, nothing to add)
                    sum += tl;
                    if (!(sum <= bsf)) {  // above eps^2, or NaN: the engine does not report it, the executor already has
                        report = true;
                        state = 2;
                    }
                }
            }
        }
        if (report && lane == 0) {
            const unsigned long long o = atomicAdd(&counters[KVM_CNT_ANSWERS], 1ULL);
            if ((int64_t)o < cap) {
                out_off[o] = sv.start;
                out_dist[o] = sqrt(bsf);                                               // :166, Math.sqrt(dist) with dist == bsf
            }
        }
    }
}

hipError_t kvm_launch_ucr_ed_screen(const double *series, const double2 *prefix, const kvm_query_dev &qd, const double *zqs,
                                    const int32_t *order, const kvm_tile *tilesA, int64_t nA, const kvm_tile *tilesB, int64_t nB,
                                    const unsigned long long *d_counts, kvm_surv *ties, int64_t cap, unsigned long long *counters,
                                    int num_cu, hipStream_t s) {
    const int64_t bound = d_counts ? nA : nA + nB;  // (device-built lists: both bounds are the number of trimmed tiles)
    if (bound <= 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<int64_t>((bound + 3) / 4, (int64_t)num_cu * 8);
    hipLaunchKernelGGL(kvm_ucr_ed_screen_kernel, dim3(grid), dim3(256), 0, s, series, prefix, qd, zqs, order, tilesA, nA, tilesB,
                       nB, d_counts, ties, cap, counters);
    return hipGetLastError();
}

hipError_t kvm_launch_ucr_ed_ties(const double *series, const kvm_query_dev &qd, const double *zqs, const int32_t *order,
                                  const kvm_surv *ties, const unsigned long long *d_n, int64_t max_ties, const kvm_replay_ws &ws,
                                  int32_t *out_off, double *out_dist, int64_t cap, unsigned long long *counters, int num_cu,
                                  hipStream_t s) {
    if (max_ties <= 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<int64_t>((max_ties + 3) / 4, (int64_t)num_cu * 8);
    hipLaunchKernelGGL(kvm_ucr_ed_tie_kernel, dim3(grid), dim3(256), 0, s, series, qd, zqs, order, ties, d_n, max_ties, ws, out_off,
                       out_dist, cap, counters);
    return hipGetLastError();
}

// ---- 2. UcrDtwQueryExecutor: what earlier windows left in cb1 / cb2 ---------------------------------------------------------
// upper / lower envelope of the fetched chunk at 1-based point p (as ref_env_at of kvm_dtw_ref.hip: a pure selection)
__device__ __forceinline__ void ucr_env_at(const double *__restrict__ series, const kvm_dtw_aux &ax, int r, int32_t cbeg,
                                           int32_t cend, int32_t p, double &U, double &L) {
    if (p - r >= cbeg && p + r <= cend) {
        U = ax.env_u[p - 1];
        L = ax.env_l[p - 1];
        return;
    }
    const int32_t a = max(cbeg, p - r), b = min(cend, p + r);
    double mx = series[a - 1], mn = mx;
    for (int32_t t = a + 1; t <= b; t++) {
        const double v = series[t - 1];
        mx = (v > mx) ? v : mx;
        mn = (v < mn) ? v : mn;
    }
    U = mx;
    L = mn;
}

// every run of the query queued for a full replay of its running sums (the depth kernel needs the reference's own mean /
// std of EVERY window)
__global__ __launch_bounds__(256) void kvm_ucr_mark_all_kernel(kvm_replay_ws ws) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < ws.n_runs) {
        ws.run_maxk[r] = ws.runs[r].count;
        ws.run_list[r] = (int32_t)r;
    }
    if (r == 0) *ws.run_list_n = (unsigned long long)ws.n_runs;
}

// One lane per window, in the reference's arithmetic (U/UcrDtwQueryExecutor.java:227-262): depth1 = iterations of
// lbKeoghCumulative (0: not reached), depth2 = iterations of lbKeoghDataCumulative; tmax1 / tmax2 = their maxima per tile.
__global__ __launch_bounds__(256) void kvm_ucr_dtw_depth_kernel(const double *__restrict__ series, kvm_query_dev qd, kvm_dtw_aux ax,
                                                                const kvm_tile *__restrict__ tiles, int64_t n_tiles,
                                                                kvm_replay_ws ws, int32_t *__restrict__ depth1,
                                                                int32_t *__restrict__ depth2, int32_t *__restrict__ tmax1,
                                                                int32_t *__restrict__ tmax2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;
    const kvm_tile tile = tiles[t];
    const int m = qd.m, r = qd.rho;
    const double bsf = qd.eps2;
    int d1 = 0, d2 = 0;
    if (lane < tile.count) {
        const kvm_run rn = ws.runs[tile.run];
        const int32_t start = tile.start + lane;
        const int64_t wi = rn.win_base + (start - rn.start);
        const double mean = ws.stats[wi] / m;                                       // :227
        const double std = sqrt(ws.stats[ws.stats_plane + wi] / m - mean * mean);   // :228-229
        const double *__restrict__ x = series + (start - 1);
        if (kvm_cnsm_filter(mean, std, qd.meanQ, qd.stdQ, qd.alpha, qd.beta)) {     // :236
            bool pass = true;
            if (m >= 6) pass = dref_kim(x, qd.q, m, mean, std, bsf) <= bsf;          // :239-241
            if (pass) {
                double lb = 0;
                int i = 0;
                for (; i < m && lb < bsf; i++) lb += dref_term_q((x[ax.order[i]] - mean) / std, ax.uo[i], ax.lo[i]);  // :244
                d1 = i;
                if (lb <= bsf) {                                                    // :245
                    int32_t cbeg, cend;
                    dref_chunk(ax, m, tile.run, cbeg, cend);
                    double lb2 = 0;
                    for (i = 0; i < m && lb2 < bsf; i++) {                          // :255
                        double U, L;
                        ucr_env_at(series, ax, r, cbeg, cend, start + ax.order[i], U, L);
                        lb2 += dref_term_d(ax.qo[i], U, L, mean, std);
                    }
                    d2 = i;
                }
            }
        }
        depth1[wi] = d1;
        depth2[wi] = d2;
    }
    int m1 = d1, m2 = d2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m1 = max(m1, __shfl_xor(m1, o, 64));
        m2 = max(m2, __shfl_xor(m2, o, 64));
    }
    if (lane == 0) {
        tmax1[t] = m1;
        tmax2[t] = m2;
    }
}

// One wavefront per candidate of the sequential tie path: the content the executor's cb1 / cb2 have when this window's loops
// start, for the positions its own loops will not overwrite.  Walks back over the earlier windows in scan order (= the order
// of the tile list): the nearest one whose loop got past depth C wrote the entries of depths [C, its depth), then C rises.
__global__ __launch_bounds__(64) void kvm_ucr_dtw_stale_kernel(const double *__restrict__ series, kvm_query_dev qd, kvm_dtw_aux ax,
                                                               const kvm_tile *__restrict__ tiles, kvm_replay_ws ws,
                                                               const int32_t *__restrict__ depth1,
                                                               const int32_t *__restrict__ depth2,
                                                               const int32_t *__restrict__ tmax1,
                                                               const int32_t *__restrict__ tmax2,
                                                               const kvm_cand *__restrict__ cands, int64_t n,
                                                               double *__restrict__ slots) {
    const int64_t idx = blockIdx.x;
    if (idx >= n) return;
    const int lane = threadIdx.x;
    const int m = qd.m, r = qd.rho;
    const kvm_cand c = cands[idx];
    double *__restrict__ cb1 = slots + (size_t)idx * 2 * m, *__restrict__ cb2 = cb1 + m;
    for (int k = lane; k < 2 * m; k += 64) cb1[k] = 0.0;   // `new double[M]`, U/UcrDtwQueryExecutor.java:157-161
    if (c.start < 0) return;
    const kvm_run rc = ws.runs[c.run];
    const int64_t tw = rc.tile_base + (c.start - rc.start) / 64;  // (the DTW plan cuts runs into tiles of 64)
    const int lw = (c.start - rc.start) % 64;
    const int64_t wic = rc.win_base + (c.start - rc.start);
    for (int which = 0; which < 2; which++) {
        const int32_t *__restrict__ depth = which ? depth2 : depth1, *__restrict__ tmax = which ? tmax2 : tmax1;
        double *__restrict__ cb = which ? cb2 : cb1;
        int C = depth[wic];  // the candidate's own loop overwrites the entries below its own depth
        int64_t t = tw;
        int limit = lw;      // lanes < limit of tile t are earlier windows
        while (C < m && t >= 0) {
            const kvm_tile tile = tiles[t];
            const kvm_run rn = ws.runs[tile.run];
            const bool have = lane < min(limit, tile.count);
            const int32_t start = tile.start + lane;
            const int64_t wi = rn.win_base + (start - rn.start);
            const int d = have ? depth[wi] : 0;
            // a record: deeper than everything between it and the candidate (and than C)
            int sfx = d;  // inclusive maximum over lanes >= mine
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_down(sfx, o, 64);
                if (lane + o < 64) sfx = max(sfx, v);
            }
            const int above = __shfl_down(sfx, 1, 64);  // exclusive: lanes > mine
            const int nearer = lane < 63 ? above : 0;
            unsigned long long rec = __ballot(d > max(C, nearer));
            while (rec && C < m) {
                const int L = 63 - __clzll((long long)rec);  // nearest record first (wave-uniform)
                rec &= ~(1ULL << L);
                const int dL = __shfl(d, L, 64);
                if (dL <= C) continue;
                const int32_t sL = tile.start + L;
                const int64_t wL = rn.win_base + (sL - rn.start);
                const double mean = ws.stats[wL] / m;
                const double std = sqrt(ws.stats[ws.stats_plane + wL] / m - mean * mean);
                const double *__restrict__ x = series + (sL - 1);
                int32_t cbeg = 0, cend = 0;
                if (which) dref_chunk(ax, m, tile.run, cbeg, cend);
                for (int i = C + lane; i < dL; i += 64) {
                    const int o = ax.order[i];
                    double term;
                    if (!which) {
                        term = dref_term_q((x[o] - mean) / std, ax.uo[i], ax.lo[i]);
                    } else {
                        double U, Lo;
                        ucr_env_at(series, ax, r, cbeg, cend, sL + o, U, Lo);
                        term = dref_term_d(ax.qo[i], U, Lo, mean, std);
                    }
                    cb[o] = term;
                }
                C = dL;
            }
            if (C >= m) break;
            // the nearest earlier tile that holds a deeper window
            limit = 64;
            int64_t tn = -1;
            for (int64_t base = t - 1; base >= 0 && tn < 0; base -= 64) {
                const int64_t tj = base - lane;
                const bool hit = tj >= 0 && tmax[tj] > C;
                const unsigned long long b = __ballot(hit);
                if (b) tn = base - (__ffsll((long long)b) - 1);
            }
            t = tn;
        }
    }
}

hipError_t kvm_launch_ucr_dtw_stale(const double *series, const kvm_query_dev &qd, const kvm_dtw_aux &ax, const kvm_tile *tiles,
                                    int64_t n_tiles, const kvm_replay_ws &ws, int32_t *depth1, int32_t *depth2, int32_t *tmax1,
                                    int32_t *tmax2, const kvm_cand *cands, int64_t n, double *slots, bool tables_ready,
                                    hipStream_t s) {
    if (n <= 0 || n_tiles <= 0) return hipSuccess;
    if (!tables_ready) {
        hipLaunchKernelGGL(kvm_ucr_mark_all_kernel, dim3((unsigned)((ws.n_runs + 255) / 256)), dim3(256), 0, s, ws);
        hipError_t e = kvm_launch_replay_runs(series, qd.m, ws, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kvm_ucr_dtw_depth_kernel, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, s, series, qd, ax, tiles,
                           n_tiles, ws, depth1, depth2, tmax1, tmax2);
    }
    hipLaunchKernelGGL(kvm_ucr_dtw_stale_kernel, dim3((unsigned)n), dim3(64), 0, s, series, qd, ax, tiles, ws, depth1, depth2, tmax1,
                       tmax2, cands, n, slots);
    return hipGetLastError();
}
