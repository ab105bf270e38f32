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
    if (alive) {
        if (NORM) {
            kvm_window_mean_std(prefix, series, (int64_t)start - 1, m, mean, std);
            pass_filter = kvm_cnsm_filter(mean, std, qd.meanQ, qd.stdQ, qd.alpha, qd.beta);
        } else {
            pass_filter = true;
        }
    }
    if (pass_filter) {
        if (use_lb && m >= 6) {
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
            pass_kim = lb <= qd.eps2;
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
        c.pad = 0;
        c.mean = mean;
        c.std = std;
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
// Stage 2: one wavefront per survivor.  LB_Keogh (query envelope, K/utils/DtwUtils.java:206-222) and
// LB_Keogh on the data envelope (K/utils/DtwUtils.java:238-257), both summed over all m positions in
// lane-strided order (the reference's order[] permutation and early exit only change the summation
// order / skip work).  A candidate survives iff both bounds are <= eps^2.
__global__ __launch_bounds__(256) void kvm_dtw_keogh_kernel(const double *__restrict__ series,
                                                            const double *__restrict__ env_u,
                                                            const double *__restrict__ env_l, kvm_query_dev qd,
                                                            const kvm_cand *__restrict__ in,
                                                            const unsigned long long *__restrict__ d_n_in,
                                                            kvm_cand *__restrict__ out,
                                                            unsigned long long *__restrict__ counters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long n_in = *d_n_in;
    const int m = qd.m;
    const double eps2 = qd.eps2;
    for (unsigned long long idx = (unsigned long long)blockIdx.x * 4 + wave; idx < n_in;
         idx += (unsigned long long)gridDim.x * 4) {
        const kvm_cand c = in[idx];
        const double *__restrict__ x = series + (c.start - 1);
        const double inv = 1.0 / c.std;
        double lb1 = 0.0;
#pragma unroll 4
        for (int j = lane; j < m; j += 64) {
            const double v = (x[j] - c.mean) * inv;
            const double hi = qd.qu[j], lo = qd.ql[j];
            double d = 0.0;
            if (v > hi) d = dsq(v, hi);
            else if (v < lo) d = dsq(v, lo);
            lb1 += d;
        }
        lb1 = kvm_wave_sum(lb1);
        if (!(lb1 <= eps2)) continue;
        bool pass2 = true;
        if (env_u) {
            const double *__restrict__ uu = env_u + (c.start - 1);
            const double *__restrict__ ll = env_l + (c.start - 1);
            double lb2 = 0.0;
#pragma unroll 4
            for (int j = lane; j < m; j += 64) {
                const double hi = (uu[j] - c.mean) * inv, lo = (ll[j] - c.mean) * inv;
                const double qv = qd.q[j];
                double d = 0.0;
                if (qv > hi) d = dsq(qv, hi);
                else if (qv < lo) d = dsq(qv, lo);
                lb2 += d;
            }
            lb2 = kvm_wave_sum(lb2);
            pass2 = lb2 <= eps2;
        }
        if (lane == 0) {
            atomicAdd(&counters[KVM_CNT_PASS_KEOGH], 1ULL);
            if (pass2) {
                atomicAdd(&counters[KVM_CNT_PASS_KEOGH2], 1ULL);
                unsigned long long o = atomicAdd(&counters[KVM_CNT_CANDS2], 1ULL);
                out[o] = c;
            }
        }
    }
}

hipError_t kvm_launch_dtw_keogh(const double *series, const double *env_u, const double *env_l, int64_t n,
                                const kvm_query_dev &qd, const kvm_cand *in, const unsigned long long *d_n_in,
                                int64_t max_in, kvm_cand *out, unsigned long long *counters, hipStream_t s) {
    (void)n;
    if (max_in <= 0) return hipSuccess;
    int64_t blocks = (max_in + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(kvm_dtw_keogh_kernel, dim3((unsigned)blocks), dim3(256), 0, s, series, env_u, env_l, qd, in,
                       d_n_in, out, counters);
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

template <bool ABANDON>
__global__ __launch_bounds__(64) void kvm_dtw_dp_lds_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                            const kvm_cand *__restrict__ in,
                                                            const unsigned long long *__restrict__ d_n_in,
                                                            int32_t *__restrict__ out_off,
                                                            double *__restrict__ out_dist, int64_t cap,
                                                            unsigned long long *__restrict__ counters) {
    const int lane = threadIdx.x;
    const int m = qd.m, r = qd.rho;
    const double eps2 = qd.eps2;
    const double *__restrict__ B = qd.q;
    double *E = kvm_dp_lds;      // r+1 entries
    double *O = E + (r + 1);     // r+1 entries (only r used)
    double *A = O + (r + 1);     // m entries: the (z-normalised) candidate window
    const unsigned long long n_in = *d_n_in;

    for (;;) {
        unsigned long long idx = 0;
        if (lane == 0) idx = atomicAdd(&counters[KVM_CNT_QUEUE], 1ULL);
        idx = __shfl(idx, 0, 64);
        if (idx >= n_in) break;
        const kvm_cand c = in[idx];
        const double *__restrict__ x = series + (c.start - 1);
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
                if (mn > eps2) {
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

hipError_t kvm_launch_dtw_dp(const double *series, int64_t n, const kvm_query_dev &qd, const kvm_cand *in,
                             const unsigned long long *d_n_in, int64_t max_in, int32_t *out_off, double *out_dist,
                             int64_t cap, unsigned long long *counters, int num_cu, hipStream_t s) {
    (void)n;
    if (max_in <= 0) return hipSuccess;
    size_t lds = ((size_t)2 * (qd.rho + 1) + qd.m) * sizeof(double);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)num_cu * per_cu;
    if (blocks > max_in) blocks = max_in;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(
            qd.early_abandon ? (const void *)kvm_dtw_dp_lds_kernel<true> : (const void *)kvm_dtw_dp_lds_kernel<false>,
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (qd.early_abandon)
        hipLaunchKernelGGL((kvm_dtw_dp_lds_kernel<true>), dim3((unsigned)blocks), dim3(64), lds, s, series, qd, in,
                           d_n_in, out_off, out_dist, cap, counters);
    else
        hipLaunchKernelGGL((kvm_dtw_dp_lds_kernel<false>), dim3((unsigned)blocks), dim3(64), lds, s, series, qd, in,
                           d_n_in, out_off, out_dist, cap, counters);
    return hipGetLastError();
}
