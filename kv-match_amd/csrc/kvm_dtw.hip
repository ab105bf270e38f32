// kvm_dtw.hip -- DTW verification cascade (RSM-DTW, cNSM-DTW) for gfx950.
//
// Replaces K/QueryEngineDtw.java:401-451 and K/NormQueryEngineDtw.java:540-599 plus the static
// functions of K/utils/DtwUtils.java.  Stages (each a kernel, survivors compacted between stages):
//   1. filter  : per-window mean/std from the prefix sums, (alpha,beta) test (cNSM only), LB_Kim.
//   2. keogh   : LB_Keogh against the query envelope and against the (global) data envelope.
//   3. dp      : Sakoe-Chiba banded DP, one wavefront per surviving candidate.
// The answer set is {filter-passing candidates with banded DTW <= eps^2}: the lower bounds only prune
// (SURVEY 8a "semantic facts"), so stages 1-2 can be switched off (KVM_OPT_LB_CASCADE=0) without
// changing any answer.
//
// Every DP cell performs exactly the reference's operations, min(min(x,y),z) + (a-b)*(a-b)
// (K/utils/DtwUtils.java:316), in fp64 without contraction, so the DP value is bit-identical to the
// reference's for identical inputs regardless of the order in which cells are visited.
#include "kvm_internal.h"

#define DTW_INF 1e20  // K/utils/DtwUtils.java:24

__device__ __forceinline__ double dsq(double a, double b) { return (a - b) * (a - b); }

// ---------------------------------------------------------------------------------------------------
// Stage 1: one lane per candidate window; one wavefront per tile (<= 64 consecutive candidates).
// LB_Kim follows K/utils/DtwUtils.java:149-189 without the intermediate early returns (they only
// skip work; the full bound is >= every partial bound, and all are lower bounds of the DTW distance).
template <bool NORM>
__global__ __launch_bounds__(256) void kvm_dtw_filter_kernel(const double *__restrict__ series,
                                                             const double2 *__restrict__ prefix, kvm_query_dev qd,
                                                             const kvm_tile *__restrict__ tiles, int64_t n_tiles,
                                                             int32_t use_lb, kvm_cand *__restrict__ out,
                                                             unsigned long long *__restrict__ counters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;
    const kvm_tile tile = tiles[t];
    const int m = qd.m;
    const double *__restrict__ q = qd.q;
    bool alive = lane < tile.count;
    bool pass_filter = false, pass_kim = false;
    const int32_t start = tile.start + lane;
    double mean = 0.0, std = 1.0;
    double lim = qd.eps2 * (1.0 + 1e-9);  // pruning threshold of the lower bounds: eps^2 plus the guard band
    if (alive) {
        if (NORM) {
            const kvm_stat_guard sg = kvm_window_mean_std_g(prefix, series, (int64_t)start - 1, m, mean, std);
            pass_filter = kvm_cnsm_filter_guarded(mean, std, qd.meanQ, qd.stdQ, qd.alpha, qd.beta, sg);
            lim = qd.eps2 + kvm_eps2_slack(qd.eps2, m, sg.g, 16.0);
            if (!(std == std) || !(std > 0.0) || kvm_guard_degenerate(sg.g)) { std = __builtin_nan(""); lim = __builtin_huge_val(); }  // exact pass decides
        } else {
            pass_filter = true;
        }
    }
    if (pass_filter) {
        if (use_lb && m >= 6 && std == std) {
            const double *__restrict__ tt = series + (start - 1);
            double d, lb;
            double x0 = (tt[0] - mean) / std;
            double y0 = (tt[m - 1] - mean) / std;
            lb = dsq(x0, q[0]) + dsq(y0, q[m - 1]);
            double x1 = (tt[1] - mean) / std;
            d = fmin(dsq(x1, q[0]), dsq(x0, q[1]));
            d = fmin(d, dsq(x1, q[1]));
            lb += d;
            double y1 = (tt[m - 2] - mean) / std;
            d = fmin(dsq(y1, q[m - 1]), dsq(y0, q[m - 2]));
            d = fmin(d, dsq(y1, q[m - 2]));
            lb += d;
            double x2 = (tt[2] - mean) / std;
            d = fmin(dsq(x0, q[2]), dsq(x1, q[2]));
            d = fmin(d, dsq(x2, q[2]));
            d = fmin(d, dsq(x2, q[1]));
            d = fmin(d, dsq(x2, q[0]));
            lb += d;
            double y2 = (tt[m - 3] - mean) / std;
            d = fmin(dsq(y0, q[m - 3]), dsq(y1, q[m - 3]));
            d = fmin(d, dsq(y2, q[m - 3]));
            d = fmin(d, dsq(y2, q[m - 2]));
            d = fmin(d, dsq(y2, q[m - 1]));
            lb += d;
            pass_kim = lb <= lim;
        } else {
            pass_kim = true;
        }
    }
    // wave-aggregated append
    const unsigned long long bf = __ballot(pass_filter), bk = __ballot(pass_kim);
    unsigned long long base = 0;
    if (lane == 0) {
        if (NORM && bf) atomicAdd(&counters[KVM_CNT_PASS_FILTER], (unsigned long long)__popcll(bf));
        if (bk) {
            atomicAdd(&counters[KVM_CNT_PASS_KIM], (unsigned long long)__popcll(bk));
            base = atomicAdd(&counters[KVM_CNT_CANDS1], (unsigned long long)__popcll(bk));
        }
    }
    base = __shfl(base, 0, 64);
    if (pass_kim) {
        const unsigned long long below = bk & ((1ULL << lane) - 1ULL);
        kvm_cand c;
        c.start = start;
        c.run_start = tile.run_start;
        c.mean = mean;
        c.std = std;
        c.lim = lim;
        out[base + __popcll(below)] = c;
    }
}

hipError_t kvm_launch_dtw_filter(const double *series, const double2 *prefix, int64_t n, const kvm_query_dev &qd,
                                 const kvm_tile *tiles, int64_t n_tiles, int32_t use_lb, kvm_cand *out,
                                 unsigned long long *counters, hipStream_t s) {
    (void)n;
    if (n_tiles <= 0) return hipSuccess;
    dim3 grid((unsigned)((n_tiles + 3) / 4)), block(256);
    if (qd.normalised)
        hipLaunchKernelGGL((kvm_dtw_filter_kernel<true>), grid, block, 0, s, series, prefix, qd, tiles, n_tiles, use_lb,
                           out, counters);
    else
        hipLaunchKernelGGL((kvm_dtw_filter_kernel<false>), grid, block, 0, s, series, prefix, qd, tiles, n_tiles,
                           use_lb, out, counters);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Stage 2: one WORKGROUP per survivor.  LB_Keogh (query envelope, K/utils/DtwUtils.java:206-222) and LB_Keogh on
// the data envelope (K/utils/DtwUtils.java:238-257), both summed over all m positions in lane-strided order (the
// reference's order[] permutation and early exit only change the summation order / skip work).  A candidate
// survives iff both bounds are <= eps^2.
//
// The per-position contributions of the larger bound feed the DP's early abandon in the reference as a suffix
// sum cb[] (K/NormQueryEngineDtw.java:576-587, used at K/utils/DtwUtils.java:324-326).  Here the positions are
// grouped into KVM_DTW_NSEG segments of seg_len points; the kernel emits the COARSE suffix sums
//     suffix[s] = sum of cb[pos] over pos >= s * seg_len
// (16 doubles per survivor instead of m), which is all the DP kernel needs because it tests for abandonment
// every 64 macro-steps anyway.  A coarse suffix is a subset sum of non-negative terms, i.e. still a lower bound.
//
// The four waves of the workgroup take four segments each.  With thousands of survivors the kernel is memory-bound
// (~6 TB/s of L2 + HBM reads) and the layout does not matter; with a few hundred (a query with few answers) it is a
// chain of global-load latencies per survivor, which this cuts by four (80 -> 25 us).
__global__ __launch_bounds__(256) void kvm_dtw_keogh_kernel(const double *__restrict__ series,
                                                            const double *__restrict__ env_u,
                                                            const double *__restrict__ env_l, kvm_query_dev qd,
                                                            const kvm_cand *__restrict__ in,
                                                            const unsigned long long *__restrict__ d_n_in,
                                                            kvm_cand *__restrict__ out, double *__restrict__ out_suffix,
                                                            int32_t seg_len, unsigned long long *__restrict__ counters) {
    __shared__ double segs[2][KVM_DTW_NSEG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long n_in = *d_n_in;
    const int m = qd.m;
    const double eps2 = qd.eps2;
    for (unsigned long long idx = blockIdx.x; idx < n_in; idx += gridDim.x) {
        const kvm_cand c = in[idx];  // (the same record in every thread: all branches on it are workgroup-uniform)
        const double *__restrict__ x = series + (c.start - 1);
        const double inv = 1.0 / c.std;
        // pruning threshold: eps^2 plus the guard band that covers the sigma uncertainty (cNSM) and summation order
        const double lim = qd.normalised ? c.lim : eps2 * (1.0 + 1e-9);
        if (!(c.std == c.std)) {  // degenerate sigma in the pruning arithmetic: no bound, the exact pass decides
            if (threadIdx.x == 0) {
                atomicAdd(&counters[KVM_CNT_PASS_KEOGH], 1ULL);
                atomicAdd(&counters[KVM_CNT_PASS_KEOGH2], 1ULL);
                const unsigned long long o = atomicAdd(&counters[KVM_CNT_CANDS2], 1ULL);
                out[o] = c;
                for (int s = 0; s < KVM_DTW_NSEG; s++) out_suffix[o * KVM_DTW_NSEG + s] = 0.0;
            }
            continue;
        }
        for (int s = wave; s < KVM_DTW_NSEG; s += 4) {
            const int j1 = min(m, (s + 1) * seg_len);
            double p = 0.0;
#pragma unroll 4
            for (int j = s * seg_len + lane; j < j1; j += 64) {
                const double v = (x[j] - c.mean) * inv;
                const double hi = qd.qu[j], lo = qd.ql[j];
                double d = 0.0;
                if (v > hi) d = dsq(v, hi);
                else if (v < lo) d = dsq(v, lo);
                p += d;
            }
            p = kvm_wave_sum(p);
            if (lane == 0) segs[0][s] = p;
        }
        __syncthreads();
        double lb1 = 0.0;
#pragma unroll
        for (int s = 0; s < KVM_DTW_NSEG; s++) lb1 += segs[0][s];
        if (!(lb1 <= lim)) {
            __syncthreads();  // segs[0] is rewritten by the next survivor
            continue;
        }
        bool pass2 = true;
        double lb2 = 0.0;
        if (env_u) {
            const double *__restrict__ uu = env_u + (c.start - 1);
            const double *__restrict__ ll = env_l + (c.start - 1);
            for (int s = wave; s < KVM_DTW_NSEG; s += 4) {
                const int j1 = min(m, (s + 1) * seg_len);
                double p = 0.0;
#pragma unroll 4
                for (int j = s * seg_len + lane; j < j1; j += 64) {
                    const double hi = (uu[j] - c.mean) * inv, lo = (ll[j] - c.mean) * inv;
                    const double qv = qd.q[j];
                    double d = 0.0;
                    if (qv > hi) d = dsq(qv, hi);
                    else if (qv < lo) d = dsq(qv, lo);
                    p += d;
                }
                p = kvm_wave_sum(p);
                if (lane == 0) segs[1][s] = p;
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < KVM_DTW_NSEG; s++) lb2 += segs[1][s];
            pass2 = lb2 <= lim;
        }
        if (threadIdx.x == 0) {
            atomicAdd(&counters[KVM_CNT_PASS_KEOGH], 1ULL);
            if (pass2) {
                atomicAdd(&counters[KVM_CNT_PASS_KEOGH2], 1ULL);
                const unsigned long long o = atomicAdd(&counters[KVM_CNT_CANDS2], 1ULL);
                out[o] = c;
                // larger bound wins, as K/NormQueryEngineDtw.java:577-587; suffix over segments
                const int which = (env_u && !(lb1 > lb2)) ? 1 : 0;
                double run = 0.0;
                for (int s = KVM_DTW_NSEG - 1; s >= 0; s--) {
                    run += segs[which][s];
                    out_suffix[o * KVM_DTW_NSEG + s] = run;
                }
            }
        }
        __syncthreads();  // segs are rewritten by the next survivor
    }
}

hipError_t kvm_launch_dtw_keogh(const double *series, const double *env_u, const double *env_l, int64_t n,
                                const kvm_query_dev &qd, const kvm_cand *in, const unsigned long long *d_n_in,
                                int64_t max_in, kvm_cand *out, double *out_suffix, int32_t seg_len,
                                unsigned long long *counters, hipStream_t s) {
    (void)n;
    if (max_in <= 0) return hipSuccess;
    int64_t blocks = max_in;  // one workgroup per survivor, 8 resident per CU
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(kvm_dtw_keogh_kernel, dim3((unsigned)blocks), dim3(256), 0, s, series, env_u, env_l, qd, in,
                       d_n_in, out, out_suffix, seg_len, counters);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Stage 3 (generic LDS form): banded DP swept by anti-diagonals, one wavefront per candidate.
//
// Band coordinate k = j - i + r in [0, 2r] (the reference's index into cost[], K/utils/DtwUtils.java:290).
// On anti-diagonal d = i + j only cells with k == d + r (mod 2) exist; they depend on k-1 (y: same row)
// and k+1 (x: previous row) of diagonal d-1 and on k of diagonal d-2 (z).  Keeping the even-k cells in
// E[] and the odd-k cells in O[] therefore allows an in-place red/black sweep:
//     d+r even:  E[e] = min(min(O[e], O[e-1]), E[e]) + dist     (k = 2e)
//     d+r odd :  O[o] = min(min(E[o+1], E[o]), O[o]) + dist     (k = 2o+1)
// Cells outside the matrix are never written and stay at INF = 1e20, which is what the reference
// substitutes for them (K/utils/DtwUtils.java:297-311).
//
// Early abandon: every path crosses diagonal d or d-1, and costs only grow along a path, so
// min(E u O) is a lower bound of the final DTW value; the candidate is dropped once it exceeds eps^2.
// (The reference abandons per row with the LB_Keogh suffix sums, K/utils/DtwUtils.java:324-326; both
// tests only ever discard candidates whose DTW distance exceeds eps^2.)
extern __shared__ double kvm_dp_lds[];

// `scratch` != nullptr: the two band rows and the window do not fit the LDS (m + 2 rho > 20 000 or so) and live in a
// per-workgroup slice of global memory instead -- same code, one wavefront per workgroup, so the workgroup barrier
// (a workgroup-scope fence) orders its stores and loads through the CU's own L1.
template <bool ABANDON>
__global__ __launch_bounds__(64) void kvm_dtw_dp_lds_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                            const kvm_cand *__restrict__ in,
                                                            const double *__restrict__ suffix, int32_t seg_len,
                                                            const unsigned long long *__restrict__ d_n_in,
                                                            int32_t *__restrict__ out_off,
                                                            double *__restrict__ out_dist, int64_t cap,
                                                            unsigned long long *__restrict__ counters,
                                                            double *scratch) {
    const int lane = threadIdx.x;
    const int m = qd.m, r = qd.rho;
    const double eps2 = qd.eps2;
    const double *__restrict__ B = qd.q;
    double *E = scratch ? scratch + (size_t)blockIdx.x * ((size_t)2 * (r + 1) + m) : kvm_dp_lds;  // r+1 entries
    double *O = E + (r + 1);     // r+1 entries (only r used)
    double *A = O + (r + 1);     // m entries: the (z-normalised) candidate window
    const unsigned long long n_in = *d_n_in;

    for (;;) {
        unsigned long long idx = 0;
        if (lane == 0) idx = atomicAdd(&counters[KVM_CNT_QUEUE], 1ULL);
        idx = __shfl(idx, 0, 64);
        if (idx >= n_in) break;
        const kvm_cand c = in[idx];
        if (c.start < 0) continue;  // rejected by the exact (alpha, beta) test
        const double *__restrict__ x = series + (c.start - 1);
        const double lim_rest = qd.normalised ? eps2 + kvm_eps2_slack(eps2, m, kvm_sigma_guard(c.mean, c.std), 16.0)
                                              : eps2 * (1.0 + 1e-9);
        // zT[k] = (T[k+j] - mean) / std   (K/NormQueryEngineDtw.java:564-567); RSM: mean 0, std 1 -> exact copy
        for (int i = lane; i < m; i += 64) A[i] = (x[i] - c.mean) / c.std;
        for (int e = lane; e <= r; e += 64) {
            E[e] = DTW_INF;
            O[e] = DTW_INF;
        }
        __syncthreads();
        bool dead = false;
        for (int d = 0; d <= 2 * m - 2; d++) {
            if (((d + r) & 1) == 0) {
                for (int e = lane; e <= r; e += 64) {
                    const int i = (d + r - 2 * e) >> 1, j = d - i;
                    if (d + r - 2 * e >= 0 && i < m && j >= 0 && j < m) {
                        const double dd = dsq(A[i], B[j]);
                        double v;
                        if (i == 0 && j == 0) {
                            v = dd;
                        } else {
                            const double xx = (e <= r - 1) ? O[e] : DTW_INF;
                            const double yy = (e >= 1) ? O[e - 1] : DTW_INF;
                            v = fmin(fmin(xx, yy), E[e]) + dd;
                        }
                        E[e] = v;
                    }
                }
            } else {
                for (int o = lane; o <= r - 1; o += 64) {
                    const int i = (d + r - 2 * o - 1) >> 1, j = d - i;
                    if (d + r - 2 * o - 1 >= 0 && i < m && j >= 0 && j < m) {
                        const double dd = dsq(A[i], B[j]);
                        if (i == 0 && j == 0) {  // r odd: the origin is an odd-k cell
                            O[o] = dd;
                        } else {
                            const double xx = E[o + 1];
                            const double yy = E[o];
                            O[o] = fmin(fmin(xx, yy), O[o]) + dd;
                        }
                    }
                }
            }
            __syncthreads();
            if (ABANDON && (d & 63) == 63 && d < 2 * m - 2) {
                double mn = DTW_INF;
                for (int e = lane; e <= r; e += 64) {
                    mn = fmin(mn, E[e]);
                    if (e <= r - 1) mn = fmin(mn, O[e]);
                }
                mn = kvm_wave_min(mn);
                // rows and columns > (d + r) / 2 are untouched by every cell of the last two diagonals: their
                // LB_Keogh contributions (coarse suffix) still have to be paid (K/utils/DtwUtils.java:324-326)
                double rest = 0.0;
                if (suffix) {
                    const int sidx = ((d + r) / 2 + 1 + seg_len - 1) / seg_len;
                    rest = sidx < KVM_DTW_NSEG ? suffix[idx * KVM_DTW_NSEG + sidx] : 0.0;
                }
                // `rest` comes from the pruning arithmetic (approximate sigma, other summation order): guard band
                if (mn + rest > (rest > 0.0 ? lim_rest : eps2)) {
                    dead = true;
                    break;
                }
            }
        }
        __syncthreads();
        if (!dead && lane == 0) {
            const double v = (r & 1) ? O[(r - 1) >> 1] : E[r >> 1];  // cell (m-1, m-1): k = r
            if (v <= eps2) {
                unsigned long long o = atomicAdd(&counters[KVM_CNT_ANSWERS], 1ULL);
                if ((int64_t)o < cap) {
                    out_off[o] = c.start;
                    out_dist[o] = sqrt(v);
                }
            }
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------
// Stage 3 (register form): the same red/black anti-diagonal sweep with the whole band in VGPRs.
//
// Lane p owns V consecutive even-k cells E[pV..pV+V) and V odd-k cells O[pV..pV+V) (V = pow2 >= (r+1)/64).
// One macro-step I performs the E-update of diagonal d = 2I - r and the O-update of diagonal d + 1:
//     E[e]: row i = I - e, column j = I - r + e          O[o]: row i = I - o, column j = I - r + 1 + o
// so both updates of cell index c use the same data value a[c] = A[I - pV - c]; the query values are
// b[c] = B[I - r + pV + c] (E) and b[c+1] (O).  From one macro-step to the next both windows slide by one, so a
// lane needs ONE new A and ONE new B value per macro-step (ds_read_b64 from two wave-private LDS rings that are
// refilled 64 values at a time); the windows live in registers and rotate by static renaming (the macro-step
// loop is unrolled V times).  The only cross-lane traffic is one halo value per update: O[pV-1] from lane p-1
// (DPP wave_shr:1) and E[(p+1)V] from lane p+1 (DPP wave_shl:1).
//
// No per-cell masks:
//   * cells outside the matrix (i or j outside [0,m)) see poisoned operands (A = +1e10, B = -1e10 outside the
//     window), so their value is >= ~1e20, which is how the reference treats them (INF = 1e20, K/utils/
//     DtwUtils.java:24,297-311) -- an in-matrix cell always has a finite predecessor, so min() never selects them
//     and the in-matrix values are bit-identical to the reference's;
//   * cell (0,0) has no predecessor in the reference (cost = dist, :292-296): its own slot is seeded with 0.0 so
//     that min(INF, INF, 0.0) + dist == dist exactly;
//   * the band edge k = -1 is lane 0's left halo (DPP fill value INF); the band edge k = 2r+1 is the single cell
//     O[r], which is reset to INF after every O-update (one v_cndmask pair).  Cells beyond it are never read
//     by an in-band cell.
#define DTW_POS_BIG 1e10
#define DTW_NEG_BIG (-1e10)

__device__ __forceinline__ double dp_min(double a, double b) {
    // one v_min_f64; fmin() would add a canonicalising v_max_f64 per operand in IEEE mode
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double dp_from_left(double v, double fill) {   // lane i <- lane i-1; lane 0 <- fill
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dp_from_right(double v, double fill) {  // lane i <- lane i+1; lane 63 <- fill
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// LDS window buffers of the register-form DP.  The data/query values a lane needs advance by one index per macro-step,
// so they are kept in LINEAR buffers (index i at position i - low): a lane's read address is base + 8*step and the
// V steps of an unrolled block use immediate offsets.  A buffer holds the 64*V values behind the front plus
// DP_EPOCH chunks of 64 new values; after DP_EPOCH chunks the live tail is slid back to the front.
#ifndef DP_EPOCH
#define DP_EPOCH 4
#endif
#ifndef DP_BLOCKS_PER_CU
#define DP_BLOCKS_PER_CU 4
#endif
template <int V>
struct dp_buf {
    static constexpr int LIVE = 64 * V;                  // values behind the front that lanes still read
    static constexpr int LIN = LIVE + 64 * DP_EPOCH;     // doubles per buffer (V=4: 768 = 6 KiB)
};

template <int V, int CSTAR, bool ABANDON>
__global__ __launch_bounds__(256) void kvm_dtw_dp_reg_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                             const kvm_cand *__restrict__ in,
                                                             const double *__restrict__ suffix, int32_t seg_len,
                                                             const unsigned long long *__restrict__ d_n_in,
                                                             int32_t *__restrict__ out_off,
                                                             double *__restrict__ out_dist, int64_t cap,
                                                             unsigned long long *__restrict__ counters) {
    constexpr int LIN = dp_buf<V>::LIN, LIVE = dp_buf<V>::LIVE;
    constexpr int WAVE_LDS = 2 * LIN + 64 * V;  // buffer A, buffer B, result scratch
    __shared__ double lds_all[4 * WAVE_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *bufA = lds_all + wave * WAVE_LDS, *bufB = bufA + LIN, *scratch = bufB + LIN;
    const int m = qd.m, r = qd.rho;
    const double eps2 = qd.eps2;
    const double *__restrict__ Bq = qd.q;
    const unsigned long long n_in = *d_n_in;
    const int I0 = r >> 1, Ilast = m - 1 + I0;
    const int pV = lane * V;
    const bool edge_lane = lane == r / V;  // the band-edge cell O[r] is cell CSTAR (= r % V) of this lane
    const int kfin = (r & 1) ? ((r - 1) >> 1) : (r >> 1);  // index of cell (m-1, m-1) (k = r) in O (r odd) / E (r even)

    for (;;) {
        unsigned long long idx = 0;
        if (lane == 0) idx = atomicAdd(&counters[KVM_CNT_QUEUE], 1ULL);
        idx = __shfl(idx, 0, 64);
        if (idx >= n_in) break;
        const kvm_cand cand = in[idx];
        if (cand.start < 0) continue;  // rejected by the exact (alpha, beta) test
        const double *__restrict__ x = series + (cand.start - 1);
        const double mean = cand.mean, sd = cand.std;
        const double lim_rest = qd.normalised ? eps2 + kvm_eps2_slack(eps2, m, kvm_sigma_guard(mean, sd), 16.0)
                                              : eps2 * (1.0 + 1e-9);
        // buffer position p holds A[lowA + p] / B[lowB + p]
        int lowA = I0 - LIVE, lowB = I0 - r;
        // zT[k] = (T[k+j] - mean) / std (K/NormQueryEngineDtw.java:564-567); RSM: (x - 0) / 1 == x exactly
        auto fillA = [&](int pos, int cnt) {
            for (int t = lane; t < cnt; t += 64) {
                const int i = lowA + pos + t;
                bufA[pos + t] = (i >= 0 && i < m) ? (x[i] - mean) / sd : DTW_POS_BIG;
            }
        };
        auto fillB = [&](int pos, int cnt) {
            for (int t = lane; t < cnt; t += 64) {
                const int j = lowB + pos + t;
                bufB[pos + t] = (j >= 0 && j < m) ? Bq[j] : DTW_NEG_BIG;
            }
        };
        fillA(0, LIVE + 64);
        fillB(0, LIVE + 64);

        double E[V], O[V], a[V], b[V];
#pragma unroll
        for (int c = 0; c < V; c++) {
            E[c] = DTW_INF;
            O[c] = DTW_INF;
            a[c] = c >= 1 ? bufA[I0 - pV - c - lowA] : 0.0;
            b[c] = bufB[I0 - r + pV + c - lowB];
        }
        // seed of cell (0,0): k = r
        if ((r & 1) == 0) {
#pragma unroll
            for (int c = 0; c < V; c++)
                if (pV + c == (r >> 1)) E[c] = 0.0;
        } else {
#pragma unroll
            for (int c = 0; c < V; c++)
                if (pV + c == ((r - 1) >> 1)) O[c] = 0.0;
        }
        // read pointers: macro-step I reads A[I - pV] and B[I - r + pV + V]
        const double *pA = bufA + (I0 - pV - lowA);
        const double *pB = bufB + (I0 - r + pV + V - lowB);
        // halos: lane 0 (left) / lane 63 (right) never receive from a neighbour, so they keep this initial INF for ever
        double halo_l = DTW_INF, halo_r = DTW_INF;
        bool dead = false, done = false;
        double result = DTW_INF;
        int chunk = 0;

        for (int Ic = I0; Ic <= Ilast && !done; Ic += 64) {
#pragma unroll 1
            for (int blk = 0; blk < 64 / V && !done; blk++) {
#pragma unroll
                for (int s = 0; s < V; s++) {
                    const int I = Ic + blk * V + s;
                    const double a_new = pA[s], b_new = pB[s];
                    // ---- E-update (diagonal 2I - r) ----
                    halo_l = dp_from_left(O[V - 1], halo_l);
#pragma unroll
                    for (int c = V - 1; c >= 0; c--) {
                        if (c == 0) a[(V - s) % V] = a_new;  // slot of cell 0 at step s
                        const double av = a[((c - s) % V + V) % V], bv = b[(c + s) % V];
                        const double dd = (av - bv) * (av - bv);
                        const double yy = c > 0 ? O[c - 1] : halo_l;
                        E[c] = dp_min(dp_min(O[c], yy), E[c]) + dd;
                    }
                    if ((r & 1) == 0 && I == Ilast) {  // r even: cell (m-1, m-1) is E[r/2] of this update
#pragma unroll
                        for (int c = 0; c < V; c++) scratch[pV + c] = E[c];
                        result = scratch[kfin];
                        done = true;
                    }
                    // ---- O-update (diagonal 2I - r + 1) ----
                    halo_r = dp_from_right(E[0], halo_r);
                    b[s % V] = b_new;  // slot (V + s) % V: query value of O cell V-1, E cell V-1 of the next step
#pragma unroll
                    for (int c = 0; c < V; c++) {
                        const double av = a[((c - s) % V + V) % V], bv = b[(c + 1 + s) % V];
                        const double dd = (av - bv) * (av - bv);
                        const double xx = c < V - 1 ? E[c + 1] : halo_r;
                        O[c] = dp_min(dp_min(xx, E[c]), O[c]) + dd;
                    }
                    if ((r & 1) == 1 && I == Ilast) {
#pragma unroll
                        for (int c = 0; c < V; c++) scratch[pV + c] = O[c];
                        result = scratch[kfin];
                        done = true;
                    }
                    // band edge k = 2r+1: O[r] must read as INF
                    O[CSTAR] = edge_lane ? DTW_INF : O[CSTAR];
                }
                pA += V;
                pB += V;
            }
            if (done) break;
            chunk++;
            if (chunk == DP_EPOCH) {
                // slide: the LIVE values behind the front move to the start of the buffers
                double ta[V], tb[V];
#pragma unroll
                for (int c = 0; c < V; c++) {
                    ta[c] = bufA[64 * DP_EPOCH + lane + 64 * c];
                    tb[c] = bufB[64 * DP_EPOCH + lane + 64 * c];
                }
#pragma unroll
                for (int c = 0; c < V; c++) {
                    bufA[lane + 64 * c] = ta[c];
                    bufB[lane + 64 * c] = tb[c];
                }
                lowA += 64 * DP_EPOCH;
                lowB += 64 * DP_EPOCH;
                pA -= 64 * DP_EPOCH;
                pB -= 64 * DP_EPOCH;
                chunk = 0;
            }
            // the 64 values the next chunk adds at the front of each window
            fillA(LIVE + 64 * chunk, 64);
            fillB(LIVE + 64 * chunk, 64);
            if (ABANDON) {
                double mn = DTW_INF;
#pragma unroll
                for (int c = 0; c < V; c++) mn = dp_min(mn, dp_min(E[c], O[c]));
                mn = kvm_wave_min(mn);
                // Every path crosses one of the last two diagonals (DTW >= mn); all their cells have row and column
                // <= Ic+63, so the LB_Keogh contributions of the positions beyond are still to be paid
                // (K/utils/DtwUtils.java:324-326 with the coarse suffix of stage 2).
                double rest = 0.0;
                if (suffix) {
                    const int sidx = (Ic + 64 + seg_len - 1) / seg_len;
                    rest = sidx < KVM_DTW_NSEG ? suffix[idx * KVM_DTW_NSEG + sidx] : 0.0;
                }
                // `rest` comes from the pruning arithmetic (approximate sigma, other summation order): guard band
                if (mn + rest > (rest > 0.0 ? lim_rest : eps2)) {
                    dead = true;
                    break;
                }
            }
        }
        if (!dead && lane == 0 && result <= eps2) {
            unsigned long long o = atomicAdd(&counters[KVM_CNT_ANSWERS], 1ULL);
            if ((int64_t)o < cap) {
                out_off[o] = cand.start;
                out_dist[o] = sqrt(result);
            }
        }
    }
}

template <int V, int CSTAR>
static hipError_t launch_dp_reg_c(const double *series, const kvm_query_dev &qd, const kvm_cand *in,
                                  const double *suffix, int32_t seg_len, const unsigned long long *d_n_in,
                                  int64_t max_in, int32_t *out_off, double *out_dist, int64_t cap,
                                  unsigned long long *counters, int num_cu, hipStream_t s) {
    // waves per SIMD = DP_BLOCKS_PER_CU (bounded by LDS: 4 waves x (2*LIN + 64V) doubles per workgroup)
    const int lds_limit = (int)((160 * 1024) / (4 * (2 * dp_buf<V>::LIN + 64 * V) * sizeof(double)));
    int64_t blocks = (int64_t)num_cu * (DP_BLOCKS_PER_CU < lds_limit ? DP_BLOCKS_PER_CU : lds_limit);
    if (blocks > (max_in + 3) / 4) blocks = (max_in + 3) / 4;
    if (qd.early_abandon)
        hipLaunchKernelGGL((kvm_dtw_dp_reg_kernel<V, CSTAR, true>), dim3((unsigned)blocks), dim3(256), 0, s, series, qd,
                           in, suffix, seg_len, d_n_in, out_off, out_dist, cap, counters);
    else
        hipLaunchKernelGGL((kvm_dtw_dp_reg_kernel<V, CSTAR, false>), dim3((unsigned)blocks), dim3(256), 0, s, series, qd,
                           in, suffix, seg_len, d_n_in, out_off, out_dist, cap, counters);
    return hipGetLastError();
}

// dispatch on the (launch-constant) position CSTAR = rho % V of the band-edge cell
template <int V>
static hipError_t launch_dp_reg(const double *series, const kvm_query_dev &qd, const kvm_cand *in,
                                const double *suffix, int32_t seg_len, const unsigned long long *d_n_in,
                                int64_t max_in, int32_t *out_off, double *out_dist, int64_t cap,
                                unsigned long long *counters, int num_cu, hipStream_t s) {
#define KVM_DP_CASE(C)                                                                                          \
    case C:                                                                                                     \
        if constexpr (C < V)                                                                                    \
            return launch_dp_reg_c<V, C>(series, qd, in, suffix, seg_len, d_n_in, max_in, out_off, out_dist, cap, \
                                         counters, num_cu, s);                                                  \
        break;
    switch (qd.rho % V) {
        KVM_DP_CASE(0) KVM_DP_CASE(1) KVM_DP_CASE(2) KVM_DP_CASE(3)
        KVM_DP_CASE(4) KVM_DP_CASE(5) KVM_DP_CASE(6) KVM_DP_CASE(7)
    }
#undef KVM_DP_CASE
    return hipErrorInvalidValue;
}

hipError_t kvm_launch_dtw_dp(const double *series, int64_t n, const kvm_query_dev &qd, const kvm_cand *in,
                             const double *suffix, int32_t seg_len, const unsigned long long *d_n_in, int64_t max_in, int32_t *out_off, double *out_dist,
                             int64_t cap, unsigned long long *counters, int num_cu, int force_lds, hipStream_t s) {
    (void)n;
    if (max_in <= 0) return hipSuccess;
    // register form: band of r+1 even-k cells over 64 lanes, V = pow2 cells per lane, V <= 8 (r <= 511)
    const int need = (qd.rho + 1 + 63) / 64;
    if (!force_lds && need <= 8 && qd.m >= 2) {
        if (need <= 1) return launch_dp_reg<1>(series, qd, in, suffix, seg_len, d_n_in, max_in, out_off, out_dist, cap, counters, num_cu, s);
        if (need <= 2) return launch_dp_reg<2>(series, qd, in, suffix, seg_len, d_n_in, max_in, out_off, out_dist, cap, counters, num_cu, s);
        if (need <= 4) return launch_dp_reg<4>(series, qd, in, suffix, seg_len, d_n_in, max_in, out_off, out_dist, cap, counters, num_cu, s);
        return launch_dp_reg<8>(series, qd, in, suffix, seg_len, d_n_in, max_in, out_off, out_dist, cap, counters, num_cu, s);
    }
    // generic form (any r): band rows + window in LDS when they fit, else in a stream-ordered global scratch
    const size_t need_bytes = ((size_t)2 * (qd.rho + 1) + qd.m) * sizeof(double);
    const bool in_lds = need_bytes <= 160 * 1024;
    const size_t lds = in_lds ? need_bytes : 0;
    int per_cu = in_lds ? (int)((160 * 1024) / need_bytes) : 4;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)num_cu * per_cu;
    if (blocks > max_in) blocks = max_in;
    double *scratch = nullptr;
    if (!in_lds) {
        hipError_t e = hipMallocAsync((void **)&scratch, need_bytes * (size_t)blocks, s);
        if (e != hipSuccess) return e;
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(
            qd.early_abandon ? (const void *)kvm_dtw_dp_lds_kernel<true> : (const void *)kvm_dtw_dp_lds_kernel<false>,
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (qd.early_abandon)
        hipLaunchKernelGGL((kvm_dtw_dp_lds_kernel<true>), dim3((unsigned)blocks), dim3(64), lds, s, series, qd, in,
                           suffix, seg_len, d_n_in, out_off, out_dist, cap, counters, scratch);
    else
        hipLaunchKernelGGL((kvm_dtw_dp_lds_kernel<false>), dim3((unsigned)blocks), dim3(64), lds, s, series, qd, in,
                           suffix, seg_len, d_n_in, out_off, out_dist, cap, counters, scratch);
    if (scratch) {
        const hipError_t e1 = hipGetLastError(), e2 = hipFreeAsync(scratch, s);
        return e1 != hipSuccess ? e1 : e2;
    }
    return hipGetLastError();
}
