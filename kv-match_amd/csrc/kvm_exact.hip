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
// is exact); only the additions, whose order defines the result, are executed sequentially by lane 0 from LDS,
// overlapped with the production of the next chunk by the other three waves of the workgroup.
// Early abandoning of the reference loops needs no replication: partial sums of squares only grow, so the
// completed sum decides identically.  No FMA anywhere (-ffp-contract=off).
#include "kvm_internal.h"

#define EX_THREADS 256   // wave 0 consumes (sequential additions), waves 1-3 produce (products, quotients)
#define EX_STEPS 512     // replay steps per chunk (4 doubles each)
#define EX_VALS 2048     // distance terms per chunk (1 double each)  == 4 * EX_STEPS
#define EX_BUF (4 * EX_STEPS)  // doubles per LDS buffer; two buffers per workgroup (32 KiB)

// lane 0: acc += buf[0..cnt) strictly in order.  16 values are fetched per round so that the LDS latency is paid once
// per 16 dependent additions (a dependent v_add_f64 takes ~16 cycles on gfx950).
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

// Producer side of the running-sum replay: steps [c0, c0+cnt) of the run that starts at 0-based point b; per step
// {d, d*d, o, o*o} where o is the point leaving the window after that step (T[j] of K/NormQueryEngine.java:523).
__device__ __forceinline__ void ex_produce_replay(const double *__restrict__ x, int64_t b, int64_t total, int m, int64_t c0,
                                                  int cnt, double *buf, int ptid, int pthreads) {
    for (int t = ptid; t < cnt; t += pthreads) {
        const int64_t g = c0 + t;
        const double d = x[b + g];
        const double o = (g >= m - 1 && g < total - 1) ? x[b + g - m + 1] : 0.0;
        *reinterpret_cast<double2 *>(buf + 4 * t) = make_double2(d, d * d);
        *reinterpret_cast<double2 *>(buf + 4 * t + 2) = make_double2(o, o * o);
    }
}

// Consumer side (one lane): ex += d; ex2 += d*d; [window complete and not ours] ex -= o; ex2 -= o*o, in this order
// (K/NormQueryEngine.java:497-499,523-524).
__device__ __forceinline__ void ex_consume_replay(const double *buf, int64_t total, int m, int64_t c0, int cnt,
                                                  double &ex, double &ex2) {
    int t = 0;
    const int n_add = (int)max((int64_t)0, min((int64_t)cnt, (int64_t)(m - 1) - c0));  // steps before the first window
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

// Replays the reference's running sums for the window starting at 0-based index `ws` of the run whose first point is
// 0-based index `b` (ws >= b) and returns that window's mean/std exactly as the reference has them.  Workgroup of
// EX_THREADS: waves 1-3 fill chunk c+1 into the other LDS buffer while lane 0 of wave 0 adds chunk c.
__device__ __forceinline__ void ex_replay_stats(const double *__restrict__ x, int64_t b, int64_t ws, int m, double *bufs,
                                                double *bcast, double &mean, double &std) {
    const int tid = threadIdx.x;
    const bool consumer = tid < 64;
    const int ptid = tid - 64, pthreads = EX_THREADS - 64;
    double ex = 0.0, ex2 = 0.0;
    const int64_t total = ws + m - b;  // points consumed up to and including the last point of the window
    const int nch = (int)((total + EX_STEPS - 1) / EX_STEPS);
    if (!consumer) ex_produce_replay(x, b, total, m, 0, (int)min((int64_t)EX_STEPS, total), bufs, ptid, pthreads);
    __syncthreads();
    for (int c = 0; c < nch; c++) {
        const int64_t c0 = (int64_t)c * EX_STEPS;
        if (consumer) {
            if (tid == 0) ex_consume_replay(bufs + (c & 1) * EX_BUF, total, m, c0, (int)min((int64_t)EX_STEPS, total - c0), ex, ex2);
        } else if (c + 1 < nch) {
            const int64_t n0 = c0 + EX_STEPS;
            ex_produce_replay(x, b, total, m, n0, (int)min((int64_t)EX_STEPS, total - n0), bufs + ((c + 1) & 1) * EX_BUF, ptid,
                              pthreads);
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double mu = ex / m;                     // K/NormQueryEngine.java:507
        bcast[0] = mu;
        bcast[1] = sqrt(ex2 / m - mu * mu);           // :508
    }
    __syncthreads();
    mean = bcast[0];
    std = bcast[1];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------
// Exact ED.  q_exact: RSM = the raw query; cNSM = zQ permuted by `order` (|zQ| descending, stable).
template <bool NORM>
__global__ __launch_bounds__(EX_THREADS) void kvm_exact_ed_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                                  const double *__restrict__ q_exact,
                                                                  const int32_t *__restrict__ order,
                                                                  const kvm_surv *__restrict__ surv,
                                                                  const unsigned long long *__restrict__ d_n_surv,
                                                                  int32_t *__restrict__ out_off,
                                                                  double *__restrict__ out_dist, int64_t cap,
                                                                  unsigned long long *__restrict__ counters) {
    __shared__ __attribute__((aligned(16))) double bufs[2 * EX_BUF];
    __shared__ double bcast[4];
    const int tid = threadIdx.x;
    const bool consumer = tid < 64;
    const int ptid = tid - 64, pthreads = EX_THREADS - 64;
    const int m = qd.m;
    const double eps2 = qd.eps2;
    const unsigned long long n_surv = *d_n_surv;
    for (unsigned long long idx = blockIdx.x; idx < n_surv; idx += gridDim.x) {
        const kvm_surv sv = surv[idx];
        const int64_t ws = (int64_t)sv.start - 1;
        const double *__restrict__ x = series + ws;
        double mean = 0.0, std = 1.0;
        if (NORM) {
            ex_replay_stats(series, (int64_t)sv.run_start - 1, ws, m, bufs, bcast, mean, std);
            if (!kvm_cnsm_filter(mean, std, qd.meanQ, qd.stdQ, qd.alpha, qd.beta)) continue;  // workgroup-uniform
            if (tid == 0) atomicAdd(&counters[KVM_CNT_PASS_FILTER_EXACT], 1ULL);
        }
        // distance terms: produced by waves 1-3 one chunk ahead, added in order by lane 0
        auto produce = [&](int c0, double *buf) {
            const int cnt = min(EX_VALS, m - c0);
            for (int t = ptid; t < cnt; t += pthreads) {
                const int k = c0 + t;
                if (NORM) {
                    const double xv = (x[order[k]] - mean) / std;  // K/NormQueryEngine.java:515
                    buf[t] = (xv - q_exact[k]) * (xv - q_exact[k]);
                } else {
                    buf[t] = (x[k] - q_exact[k]) * (x[k] - q_exact[k]);  // K/QueryEngine.java:357
                }
            }
        };
        double dist = 0.0;
        const int nch = (m + EX_VALS - 1) / EX_VALS;
        if (!consumer) produce(0, bufs);
        __syncthreads();
        for (int c = 0; c < nch; c++) {
            const int c0 = c * EX_VALS;
            if (consumer) {
                if (tid == 0) {
                    dist = ex_seq_add(bufs + (c & 1) * EX_BUF, min(EX_VALS, m - c0), dist);
                    bcast[2] = dist;
                }
            } else if (c + 1 < nch) {
                produce(c0 + EX_VALS, bufs + ((c + 1) & 1) * EX_BUF);
            }
            __syncthreads();
            const bool over = !(bcast[2] <= eps2);  // the reference's loop guard: partial sums only grow
            __syncthreads();
            if (over) break;
        }
        if (tid == 0) {
            dist = bcast[2];
            atomicAdd(&counters[KVM_CNT_RECHECKED], 1ULL);
            if (dist <= eps2) {
                unsigned long long o = atomicAdd(&counters[KVM_CNT_ANSWERS], 1ULL);
                if ((int64_t)o < cap) {
                    out_off[o] = sv.start;
                    out_dist[o] = sqrt(dist);
                }
            }
        }
        __syncthreads();
    }
}

hipError_t kvm_launch_exact_ed(const double *series, const kvm_query_dev &qd, const double *q_exact,
                               const int32_t *order, const kvm_surv *surv, const unsigned long long *d_n_surv,
                               int64_t max_surv, int32_t *out_off, double *out_dist, int64_t cap,
                               unsigned long long *counters, int num_cu, hipStream_t s) {
    if (max_surv <= 0) return hipSuccess;
    int64_t blocks = std::min<int64_t>(max_surv, (int64_t)num_cu * 5);  // 32 KiB LDS per workgroup: 5 resident per CU
    if (qd.normalised)
        hipLaunchKernelGGL((kvm_exact_ed_kernel<true>), dim3((unsigned)blocks), dim3(EX_THREADS), 0, s, series, qd,
                           q_exact, order, surv, d_n_surv, out_off, out_dist, cap, counters);
    else
        hipLaunchKernelGGL((kvm_exact_ed_kernel<false>), dim3((unsigned)blocks), dim3(EX_THREADS), 0, s, series, qd,
                           q_exact, order, surv, d_n_surv, out_off, out_dist, cap, counters);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Exact mean/std for the DTW candidates that reached the DP stage (cNSM-DTW): replay, exact (alpha, beta) test,
// records updated in place; candidates failing the exact test are marked with start = -start (the DP kernel skips them).
__global__ __launch_bounds__(EX_THREADS) void kvm_exact_stats_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                                     kvm_cand *__restrict__ cands,
                                                                     const unsigned long long *__restrict__ d_n,
                                                                     unsigned long long *__restrict__ counters) {
    __shared__ __attribute__((aligned(16))) double bufs[2 * EX_BUF];
    __shared__ double bcast[4];
    const unsigned long long n = *d_n;
    for (unsigned long long idx = blockIdx.x; idx < n; idx += gridDim.x) {
        kvm_cand c = cands[idx];
        double mean, std;
        ex_replay_stats(series, (int64_t)c.run_start - 1, (int64_t)c.start - 1, qd.m, bufs, bcast, mean, std);
        const bool pass = kvm_cnsm_filter(mean, std, qd.meanQ, qd.stdQ, qd.alpha, qd.beta);
        if (threadIdx.x == 0) {
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
    int64_t blocks = std::min<int64_t>(max_n, (int64_t)num_cu * 5);
    hipLaunchKernelGGL(kvm_exact_stats_kernel, dim3((unsigned)blocks), dim3(EX_THREADS), 0, s, series, qd, cands, d_n,
                       counters);
    return hipGetLastError();
}
