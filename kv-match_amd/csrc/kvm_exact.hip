// kvm_exact.hip -- the exact final pass: every survivor of the fast kernels is re-evaluated in the REFERENCE'S OWN
// OPERATION ORDER, so that answer membership and the reported distances are bit-identical to the Java loops:
//
//   * RSM-ED   : dist = sum_j (x[j]-q[j])*(x[j]-q[j]) added sequentially, j = 0..m-1 (K/QueryEngine.java:355-358);
//   * cNSM     : mean/std from the per-interval running sums ex += d, ex2 += d*d, with the departing point subtracted
//                after each window (K/NormQueryEngine.java:487,497-509,523-524) -- replayed from the first point
//                of the candidate's interval, so even the rounding drift of the reference is reproduced;
//                (alpha, beta) test exactly as :511; distance over the |zQ|-descending permutation with a true division
//                per point (:443-452, :514-517);
//   * DTW      : the same replayed mean/std feed the DP kernel, whose cells are bit-exact by construction.
//
// The fast kernels (prefix-sum mu/sigma, FMA, tree reductions) are therefore only PRUNERS: they keep every
// candidate whose decision quantity is within a guard band of a threshold, and this pass decides.  Products and
// quotients are computed by all 64 lanes (each is an independently rounded fp64 operation, so parallel evaluation
// is exact); only the additions, whose order defines the result, are executed sequentially by lane 0 from LDS.
// Early abandoning of the reference loops needs no replication: partial sums of squares only grow, so the
// completed sum decides identically.  No FMA anywhere (-ffp-contract=off).
#include "kvm_internal.h"

#define EX_CH 2048  // values per LDS chunk

// lane 0: acc += buf[0..cnt) strictly in order.  16 values are fetched per round so that the LDS latency is paid once
// per 16 dependent additions.
__device__ __forceinline__ double ex_seq_add(const double *buf, int cnt, double acc) {
    int t = 0;
    for (; t + 16 <= cnt; t += 16) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const double2 *>(buf + t + 2 * u);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc += v[u].x;
            acc += v[u].y;
        }
    }
    for (; t < cnt; t++) acc += buf[t];
    return acc;
}

// Replays the reference's running sums for the window starting at 0-based index `ws` of the run whose first point is
// 0-based index `b` (ws >= b), and returns mean/std of that window exactly as the reference would have them.
// One wavefront (= one workgroup of 64 threads).  buf holds EX_CH steps x {d, d*d, o, o*o}: the products are formed
// by all lanes (each an independently rounded operation), lane 0 only performs the additions, in the reference's
// order: ex += d; ex2 += d*d; [window complete] ex -= o; ex2 -= o*o  (K/NormQueryEngine.java:497-499,523-524).
__device__ __forceinline__ void ex_replay_stats(const double *__restrict__ x, int64_t b, int64_t ws, int m, int lane,
                                                double *buf, double &mean, double &std) {
    double ex = 0.0, ex2 = 0.0;
    const int64_t total = ws + m - b;  // points consumed up to and including the last point of the window
    constexpr int STEPS = EX_CH / 2;   // 4 doubles per step in a 2*EX_CH buffer
    for (int64_t c0 = 0; c0 < total; c0 += STEPS) {
        const int cnt = (int)min((int64_t)STEPS, total - c0);
        for (int t = lane; t < cnt; t += 64) {
            const int64_t g = c0 + t;  // step index; point index b + g
            const double d = x[b + g];
            // the point leaving the window after step g (T[j] of :523): only for the windows before ours
            const double o = (g >= m - 1 && g < total - 1) ? x[b + g - m + 1] : 0.0;
            *reinterpret_cast<double2 *>(buf + 4 * t) = make_double2(d, d * d);
            *reinterpret_cast<double2 *>(buf + 4 * t + 2) = make_double2(o, o * o);
        }
        __syncthreads();
        if (lane == 0) {
            int t = 0;
            // steps before the first complete window: additions only
            const int n_add = (int)max((int64_t)0, min((int64_t)cnt, (int64_t)(m - 1) - c0));
            for (; t + 4 <= n_add; t += 4) {
                double2 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const double2 *>(buf + 4 * (t + u));
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    ex += v[u].x;
                    ex2 += v[u].y;
                }
            }
            for (; t < n_add; t++) {
                ex += buf[4 * t];
                ex2 += buf[4 * t + 1];
            }
            for (; t < cnt; t++) {
                const int64_t g = c0 + t;
                const double2 dv = *reinterpret_cast<const double2 *>(buf + 4 * t);
                const double2 ov = *reinterpret_cast<const double2 *>(buf + 4 * t + 2);
                ex += dv.x;
                ex2 += dv.y;
                if (g < total - 1) {
                    ex -= ov.x;
                    ex2 -= ov.y;
                }
            }
        }
        __syncthreads();
    }
    if (lane == 0) {
        mean = ex / m;                        // K/NormQueryEngine.java:507
        std = sqrt(ex2 / m - mean * mean);    // :508
    }
    mean = __shfl(mean, 0, 64);
    std = __shfl(std, 0, 64);
}

// ---------------------------------------------------------------------------------------------------
// Exact ED.  q_exact: RSM = the raw query; cNSM = zQ permuted by `order` (|zQ| descending, stable).
template <bool NORM>
__global__ __launch_bounds__(64) void kvm_exact_ed_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                          const double *__restrict__ q_exact,
                                                          const int32_t *__restrict__ order,
                                                          const kvm_surv *__restrict__ surv,
                                                          const unsigned long long *__restrict__ d_n_surv,
                                                          int32_t *__restrict__ out_off, double *__restrict__ out_dist,
                                                          int64_t cap, unsigned long long *__restrict__ counters) {
    __shared__ __attribute__((aligned(16))) double bufA[2 * EX_CH];
    const int lane = threadIdx.x;
    const int m = qd.m;
    const double eps2 = qd.eps2;
    const unsigned long long n_surv = *d_n_surv;
    for (unsigned long long idx = blockIdx.x; idx < n_surv; idx += gridDim.x) {
        const kvm_surv sv = surv[idx];
        const int64_t ws = (int64_t)sv.start - 1;
        const double *__restrict__ x = series + ws;
        double mean = 0.0, std = 1.0;
        if (NORM) {
            ex_replay_stats(series, (int64_t)sv.run_start - 1, ws, m, lane, bufA, mean, std);
            if (!kvm_cnsm_filter(mean, std, qd.meanQ, qd.stdQ, qd.alpha, qd.beta)) continue;  // wave-uniform
            if (lane == 0) atomicAdd(&counters[KVM_CNT_PASS_FILTER_EXACT], 1ULL);
        }
        double dist = 0.0;
        bool over = false;
        for (int c0 = 0; c0 < m && !over; c0 += EX_CH) {
            const int cnt = min(EX_CH, m - c0);
            for (int t = lane; t < cnt; t += 64) {
                const int k = c0 + t;
                if (NORM) {
                    const double xv = (x[order[k]] - mean) / std;  // K/NormQueryEngine.java:515
                    bufA[t] = (xv - q_exact[k]) * (xv - q_exact[k]);
                } else {
                    bufA[t] = (x[k] - q_exact[k]) * (x[k] - q_exact[k]);  // K/QueryEngine.java:357
                }
            }
            __syncthreads();
            if (lane == 0) dist = ex_seq_add(bufA, cnt, dist);
            __syncthreads();
            dist = __shfl(dist, 0, 64);
            over = !(dist <= eps2);  // the reference's loop guard: partial sums only grow
        }
        if (lane == 0) {
            atomicAdd(&counters[KVM_CNT_RECHECKED], 1ULL);
            if (dist <= eps2) {
                unsigned long long o = atomicAdd(&counters[KVM_CNT_ANSWERS], 1ULL);
                if ((int64_t)o < cap) {
                    out_off[o] = sv.start;
                    out_dist[o] = sqrt(dist);
                }
            }
        }
    }
}

hipError_t kvm_launch_exact_ed(const double *series, const kvm_query_dev &qd, const double *q_exact,
                               const int32_t *order, const kvm_surv *surv, const unsigned long long *d_n_surv,
                               int64_t max_surv, int32_t *out_off, double *out_dist, int64_t cap,
                               unsigned long long *counters, int num_cu, hipStream_t s) {
    if (max_surv <= 0) return hipSuccess;
    int64_t blocks = std::min<int64_t>(max_surv, (int64_t)num_cu * 16);
    if (qd.normalised)
        hipLaunchKernelGGL((kvm_exact_ed_kernel<true>), dim3((unsigned)blocks), dim3(64), 0, s, series, qd, q_exact,
                           order, surv, d_n_surv, out_off, out_dist, cap, counters);
    else
        hipLaunchKernelGGL((kvm_exact_ed_kernel<false>), dim3((unsigned)blocks), dim3(64), 0, s, series, qd, q_exact,
                           order, surv, d_n_surv, out_off, out_dist, cap, counters);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Exact mean/std for the DTW candidates that reached the DP stage (cNSM-DTW): replay, exact (alpha, beta) test,
// records updated in place; candidates failing the exact test are marked with start = -start (the DP kernel skips them).
__global__ __launch_bounds__(64) void kvm_exact_stats_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                             kvm_cand *__restrict__ cands,
                                                             const unsigned long long *__restrict__ d_n,
                                                             unsigned long long *__restrict__ counters) {
    __shared__ __attribute__((aligned(16))) double bufA[2 * EX_CH];
    const int lane = threadIdx.x;
    const unsigned long long n = *d_n;
    for (unsigned long long idx = blockIdx.x; idx < n; idx += gridDim.x) {
        kvm_cand c = cands[idx];
        double mean, std;
        ex_replay_stats(series, (int64_t)c.run_start - 1, (int64_t)c.start - 1, qd.m, lane, bufA, mean, std);
        const bool pass = kvm_cnsm_filter(mean, std, qd.meanQ, qd.stdQ, qd.alpha, qd.beta);
        if (lane == 0) {
            c.mean = mean;
            c.std = std;
            if (!pass) c.start = -c.start;
            cands[idx] = c;
            atomicAdd(&counters[KVM_CNT_RECHECKED], 1ULL);
        }
    }
}

hipError_t kvm_launch_exact_stats(const double *series, const kvm_query_dev &qd, kvm_cand *cands,
                                  const unsigned long long *d_n, int64_t max_n, unsigned long long *counters,
                                  int num_cu, hipStream_t s) {
    if (max_n <= 0) return hipSuccess;
    int64_t blocks = std::min<int64_t>(max_n, (int64_t)num_cu * 16);
    hipLaunchKernelGGL(kvm_exact_stats_kernel, dim3((unsigned)blocks), dim3(64), 0, s, series, qd, cands, d_n, counters);
    return hipGetLastError();
}
