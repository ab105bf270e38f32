// kvm_dtw_lb.hip -- the screening stages of the DTW cascade for gfx950 (approximate mean/std from the prefix sums):
//   kvm_retile_dtw_kernel   (alpha, beta) filter + LB_Kim for every candidate, tiles trimmed to the survivors;
//   kvm_dtw_screen_kernel   LB_Keogh on the query envelope, summed in the reference's order with its early exit.
// Replaces, for the bulk of the candidates, K/NormQueryEngineDtw.java:550-563 / K/QueryEngineDtw.java:409-419:
//   mean/std -> |mean-meanQ| <= beta, std/stdQ in [1/alpha, alpha] -> lbKimHierarchy (K/utils/DtwUtils.java:149-189)
//   -> lbKeoghCumulative (:206-222).
// A screen may only drop a candidate that the reference certainly drops (see "DTW cascade" in kvm_internal.h): all values
// here carry the guard band of the approximate mean/std, and survivors are decided by the exact stage (kvm_dtw.hip).
#include "kvm_internal.h"

#define SCREEN_TERMS 512  // leading terms (in the reference's order) the screen looks at

__device__ __forceinline__ double lb4_dsq(double a, double b) { return (a - b) * (a - b); }

// (alpha, beta) filter + LB_Kim of one candidate with the pruning arithmetic (prefix-sum mean/std, guard bands); also
// returns the values the screen keeps per candidate.  `w0` = 0-based window start.
//
// LB_Kim may only DROP a candidate when the reference would certainly drop it.  The reference returns the first of its
// partial bounds that is >= eps^2 (after the 1st .. 4th group of terms, else the full bound) and passes the candidate on
// iff that value is <= eps^2 (K/utils/DtwUtils.java:156-188, K/NormQueryEngineDtw.java:555-556): a partial that lands
// exactly on eps^2 passes.  So: v = the first partial >= eps^2 - slack (else the full bound); drop iff v > eps^2 + slack
// -- then every earlier partial was certainly below eps^2 and this one is certainly above.  Anything else (a partial
// inside the band, NaN) is kept for the exact stage.
template <bool NORM>
__device__ __forceinline__ void lb4_prefilter(const double *__restrict__ series, const double2 *__restrict__ prefix,
                                              const kvm_query_dev &qd, int64_t w0, int32_t run, bool &pass_filter,
                                              bool &pass_kim, double &mean, double &sd, double &inv, double &nmi, double &lim) {
    const int m = qd.m;
    mean = 0.0; sd = 1.0; inv = 1.0; nmi = 0.0;
    lim = qd.eps2 * (1.0 + 1e-9);
    pass_filter = true;
    if (NORM) {
        const kvm_stat_guard sg = kvm_window_mean_std_g(prefix, series, w0, m, mean, sd, kvm_run_hist_of(qd, run));
        pass_filter = kvm_cnsm_filter_guarded(mean, sd, qd.meanQ, qd.stdQ, qd.alpha, qd.beta, sg);
        lim = qd.eps2 + kvm_eps2_slack(qd.eps2, m, sg.g, 16.0);
        if (!(sd == sd) || !(sd > 0.0) || kvm_guard_degenerate(sg.g)) {  // degenerate sigma: no bound, the exact pass decides
            sd = __builtin_nan("");
            lim = __builtin_huge_val();
            inv = 0.0;
            nmi = 0.0;
        } else {
            inv = 1.0 / sd;
            nmi = -mean * inv;
        }
    }
    pass_kim = pass_filter;
    if (pass_filter && m >= 6 && sd == sd) {
        const double *__restrict__ tt = series + w0;
        const double *__restrict__ q = qd.q;
        const double lo_thr = 2.0 * qd.eps2 - lim;  // eps^2 - slack
        double d;
        const double x0 = (tt[0] - mean) / sd, y0 = (tt[m - 1] - mean) / sd;
        const double p1 = lb4_dsq(x0, q[0]) + lb4_dsq(y0, q[m - 1]);
        const double x1 = (tt[1] - mean) / sd;
        d = fmin(lb4_dsq(x1, q[0]), lb4_dsq(x0, q[1]));
        d = fmin(d, lb4_dsq(x1, q[1]));
        const double p2 = p1 + d;
        const double y1 = (tt[m - 2] - mean) / sd;
        d = fmin(lb4_dsq(y1, q[m - 1]), lb4_dsq(y0, q[m - 2]));
        d = fmin(d, lb4_dsq(y1, q[m - 2]));
        const double p3 = p2 + d;
        const double x2 = (tt[2] - mean) / sd;
        d = fmin(lb4_dsq(x0, q[2]), lb4_dsq(x1, q[2]));
        d = fmin(d, lb4_dsq(x2, q[2]));
        d = fmin(d, lb4_dsq(x2, q[1]));
        d = fmin(d, lb4_dsq(x2, q[0]));
        const double p4 = p3 + d;
        const double y2 = (tt[m - 3] - mean) / sd;
        d = fmin(lb4_dsq(y0, q[m - 3]), lb4_dsq(y1, q[m - 3]));
        d = fmin(d, lb4_dsq(y2, q[m - 3]));
        d = fmin(d, lb4_dsq(y2, q[m - 2]));
        d = fmin(d, lb4_dsq(y2, q[m - 1]));
        const double p5 = p4 + d;
        const double v = p1 >= lo_thr ? p1 : p2 >= lo_thr ? p2 : p3 >= lo_thr ? p3 : p4 >= lo_thr ? p4 : p5;
        pass_kim = !(v > lim);
    }
}

// LB_Keogh on the query envelope as a SCREEN (lbKeoghCumulative, K/utils/DtwUtils.java:206-222): one lane per candidate
// (a wavefront = one trimmed tile of <= 64 consecutive candidates, so the window loads of a step are one coalesced
// 512-byte access), the terms added IN THE REFERENCE'S ORDER (`order[]`: identity for RSM-DTW, the reference's reorder
// for cNSM-DTW) with the pruning arithmetic (prefix-sum mean/std, v = x * inv_sigma - mean * inv_sigma, fma).  A lane
// stops at its first partial sum >= eps^2 - slack, exactly where the reference's loop `lb < bsf` stops: the candidate is
// dropped iff that partial is also > eps^2 + slack (the reference then certainly stops at the same term with a value
// certainly above eps^2); a partial inside the band -- it may equal eps^2 in the reference's arithmetic, in which case
// the reference passes the candidate on with a partially filled cb1[] -- keeps it for the exact stage, like a sum that
// has not reached the band after SCREEN_TERMS terms.  The wavefront ends when all its lanes have stopped: a few terms
// for ordinary candidates.
template <bool NORM>
__global__ __launch_bounds__(256) void kvm_dtw_screen_kernel(const double *__restrict__ series,
                                                             const double2 *__restrict__ prefix, kvm_query_dev qd,
                                                             kvm_dtw_aux ax, const kvm_tile *__restrict__ tiles,
                                                             int64_t n_tiles,
                                                             const unsigned long long *__restrict__ d_n_tiles,
                                                             kvm_cand *__restrict__ out,
                                                             unsigned long long *__restrict__ counters) {
    if (d_n_tiles) n_tiles = (int64_t)*d_n_tiles;  // tile list built on the device: the launch only knows a bound
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;
    const kvm_tile tile = tiles[t];
    const int m = qd.m;
    bool pf = false, pk = false;
    double mean = 0.0, sd = 1.0, inv = 1.0, nmi = 0.0, lim = qd.eps2 * (1.0 + 1e-9);
    const int64_t w0 = (int64_t)tile.start - 1 + (lane < tile.count ? lane : 0);
    if (lane < tile.count) lb4_prefilter<NORM>(series, prefix, qd, w0, tile.run, pf, pk, mean, sd, inv, nmi, lim);
    if (!d_n_tiles) {  // (with device-built tiles the retile kernel has counted every candidate)
        const unsigned long long bf = __ballot(pf), bk = __ballot(pk);
        if (lane == 0) {
            if (NORM && bf) atomicAdd(&counters[KVM_CNT_PASS_FILTER], (unsigned long long)__popcll(bf));
            if (bk) atomicAdd(&counters[KVM_CNT_PASS_KIM], (unsigned long long)__popcll(bk));
        }
    }
    const double lo_thr = 2.0 * qd.eps2 - lim;
    // (the reference tests `lb < bsf` before the first term too: with eps^2 - slack <= 0 nothing can be decided here)
    bool alive = pk && sd == sd && lo_thr > 0.0;
    bool rejected = false;
    double p = 0.0;
    const double *__restrict__ xw = series + w0;
    // ---- phase 1: one lane per candidate, 8 terms per round; the loads of round k+1 are in flight while round k is
    //      added up (a round is otherwise one exposed global-load latency: ~1 us, 0.5 ms for a window of 4 096 points)
    double xa[8], ua[8], la[8], xb[8], ub[8], lb_[8];  // two register sets, named statically (no indexed register arrays)
#define SCREEN_FETCH(I0, X, U, L)                                                              \
    do {                                                                                       \
        _Pragma("unroll") for (int u = 0; u < 8; u++) {                                        \
            const int i_ = min((I0) + u, m - 1); /* clamped: loads are unconditional */        \
            X[u] = xw[ax.order[i_]];                                                           \
            U[u] = ax.uo[i_];                                                                  \
            L[u] = ax.lo[i_];                                                                  \
        }                                                                                      \
    } while (0)
#define SCREEN_ROUND(I0, X, U, L)                                                              \
    do {                                                                                       \
        _Pragma("unroll") for (int u = 0; u < 8; u++) {                                        \
            const double z_ = NORM ? __builtin_fma(X[u], inv, nmi) : X[u];                     \
            double t_ = fmax(z_ - U[u], L[u] - z_);                                            \
            t_ = fmax(t_, 0.0);                                                                \
            const double pn_ = __builtin_fma(t_, t_, p);                                       \
            const bool step_ = alive && ((I0) + u < m);                                        \
            p = step_ ? pn_ : p;                                                               \
            if (step_ && p >= lo_thr) {                                                        \
                rejected = p > lim;                                                            \
                alive = false;                                                                 \
            }                                                                                  \
        }                                                                                      \
    } while (0)
    // Only the first SCREEN_TERMS terms are looked at: ordinary candidates cross eps^2 within a few terms, and a candidate
    // that is still undecided after them goes to the exact stage, whose prologue continues the same ordered sums 64
    // terms per step across a wavefront (a lane-per-candidate loop over a whole window is m / 8 exposed load latencies:
    // 0.5 ms at m = 4 096, which one near-match per query used to cost; 512 terms: at most ~60 us).
    const int m_scr = min(m, SCREEN_TERMS);
    int i0 = 0;
    if (__any(alive)) {
        SCREEN_FETCH(0, xa, ua, la);
        while (i0 < m_scr) {
            if (!__any(alive)) break;
            if (i0 + 8 < m_scr) SCREEN_FETCH(i0 + 8, xb, ub, lb_);
            SCREEN_ROUND(i0, xa, ua, la);
            i0 += 8;
            if (i0 >= m_scr || !__any(alive)) break;
            if (i0 + 8 < m_scr) SCREEN_FETCH(i0 + 8, xa, ua, la);
            SCREEN_ROUND(i0, xb, ub, lb_);
            i0 += 8;
        }
    }
#undef SCREEN_FETCH
#undef SCREEN_ROUND
    const bool keep = pk && !rejected;
    const unsigned long long bal = __ballot(keep);
    if (bal == 0ULL) return;
    unsigned long long base = 0;
    if (lane == 0) {
        atomicAdd(&counters[KVM_CNT_PASS_KEOGH], (unsigned long long)__popcll(bal));
        base = atomicAdd(&counters[KVM_CNT_CANDS1], (unsigned long long)__popcll(bal));
    }
    base = __shfl(base, 0, 64);
    if (keep) {
        kvm_cand c;
        c.start = tile.start + lane;
        c.run = tile.run;
        c.mean = mean;
        c.std = sd;
        c.lim = lim;
        out[base + __popcll(bal & ((1ULL << lane) - 1ULL))] = c;
    }
}

// DTW retile: the (alpha, beta) filter and LB_Kim for EVERY candidate, one lane each, then every 64-candidate tile is
// trimmed to the span of its survivors and appended to a device-side list for the screen (64 tiles per workgroup
// collected in LDS, one global atomic).
template <bool NORM>
__global__ __launch_bounds__(256) void kvm_retile_dtw_kernel(const double *__restrict__ series,
                                                             const double2 *__restrict__ prefix, kvm_query_dev qd,
                                                             const kvm_tile *__restrict__ tiles, int64_t n_tiles,
                                                             kvm_tile *__restrict__ out_tiles,
                                                             unsigned long long *__restrict__ counters) {
    __shared__ kvm_tile outs[64];
    __shared__ int n_out;
    __shared__ unsigned n_f, n_k;
    __shared__ unsigned long long base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        n_out = 0;
        n_f = 0;
        n_k = 0;
    }
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * 64 + wave * 16;
    unsigned cf = 0, ck = 0;
    for (int i = 0; i < 16; i++) {
        const int64_t t = t0 + i;
        if (t >= n_tiles) break;  // wave-uniform
        const kvm_tile tl = tiles[t];
        bool pf = false, pk = false;
        if (lane < tl.count) {
            double mean, sd, inv, nmi, lim;
            lb4_prefilter<NORM>(series, prefix, qd, (int64_t)tl.start - 1 + lane, tl.run, pf, pk, mean, sd, inv, nmi, lim);
        }
        const unsigned long long mf = __ballot(pf), mk = __ballot(pk);
        cf += (unsigned)__popcll(mf);
        ck += (unsigned)__popcll(mk);
        if (mk != 0 && lane == 0) {
            const int first = __ffsll((long long)mk) - 1, last = 63 - __clzll((long long)mk);
            kvm_tile o;
            o.run_start = tl.run_start;
            o.run = tl.run;
            o.start = tl.start + first;
            o.count = last - first + 1;
            outs[atomicAdd(&n_out, 1)] = o;
        }
    }
    if (lane == 0) {
        if (cf) atomicAdd(&n_f, cf);
        if (ck) atomicAdd(&n_k, ck);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        base = n_out ? atomicAdd(&counters[KVM_CNT_TILES_A], (unsigned long long)n_out) : 0;
        if (NORM && n_f) atomicAdd(&counters[KVM_CNT_PASS_FILTER], (unsigned long long)n_f);
        if (n_k) atomicAdd(&counters[KVM_CNT_PASS_KIM], (unsigned long long)n_k);
    }
    __syncthreads();
    if ((int)threadIdx.x < n_out) out_tiles[base + threadIdx.x] = outs[threadIdx.x];
}

hipError_t kvm_launch_retile_dtw(const double *series, const double2 *prefix, const kvm_query_dev &qd, const kvm_tile *tiles,
                                 int64_t n_tiles, kvm_tile *out_tiles, unsigned long long *counters, hipStream_t s) {
    if (n_tiles <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_tiles + 63) / 64));
    if (qd.normalised)
        hipLaunchKernelGGL((kvm_retile_dtw_kernel<true>), grid, dim3(256), 0, s, series, prefix, qd, tiles, n_tiles, out_tiles,
                           counters);
    else
        hipLaunchKernelGGL((kvm_retile_dtw_kernel<false>), grid, dim3(256), 0, s, series, prefix, qd, tiles, n_tiles, out_tiles,
                           counters);
    return hipGetLastError();
}

// `d_n_tiles` != nullptr: `tiles` was built on the device and holds *d_n_tiles entries; `n_tiles` is then an upper bound.
hipError_t kvm_launch_dtw_screen(const double *series, const double2 *prefix, const kvm_query_dev &qd, const kvm_dtw_aux &ax,
                                 const kvm_tile *tiles, int64_t n_tiles, const unsigned long long *d_n_tiles, kvm_cand *out,
                                 unsigned long long *counters, hipStream_t s) {
    if (n_tiles <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_tiles + 3) / 4));
    if (qd.normalised)
        hipLaunchKernelGGL((kvm_dtw_screen_kernel<true>), grid, dim3(256), 0, s, series, prefix, qd, ax, tiles, n_tiles,
                           d_n_tiles, out, counters);
    else
        hipLaunchKernelGGL((kvm_dtw_screen_kernel<false>), grid, dim3(256), 0, s, series, prefix, qd, ax, tiles, n_tiles,
                           d_n_tiles, out, counters);
    return hipGetLastError();
}
