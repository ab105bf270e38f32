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

// ---------------------------------------------------------------------------------------------------
// Running-sum replay, one workgroup per run.  Survivors of one run share the prefix of the reference's running sums,
// so the run is replayed ONCE, up to the last window any survivor needs, and {mean, std} is stored for every flagged
// window.  (Replaying per survivor would repeat the prefix for each of them.)
__device__ __forceinline__ int64_t ex_find_run(const kvm_run *__restrict__ runs, int64_t n_runs, int32_t run_start) {
    int64_t lo = 0, hi = n_runs - 1;  // runs are sorted by start and disjoint
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (runs[mid].start < run_start) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <typename REC>
__global__ __launch_bounds__(256) void kvm_replay_mark_kernel(const REC *__restrict__ recs,
                                                              const unsigned long long *__restrict__ d_n,
                                                              kvm_replay_ws ws) {
    const unsigned long long n = *d_n;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const REC rec = recs[i];
        if (rec.start < 0) continue;
        const int64_t r = ex_find_run(ws.runs, ws.n_runs, rec.run_start);
        const int32_t k = rec.start - rec.run_start;
        ws.need[ws.runs[r].win_base + k] = 1;
        atomicMax(&ws.run_maxk[r], k + 1);
    }
}

__global__ __launch_bounds__(EX_THREADS) void kvm_replay_runs_kernel(const double *__restrict__ x, int m, kvm_replay_ws ws) {
    __shared__ __attribute__((aligned(16))) double bufs[2 * EX_BUF];
    __shared__ unsigned char flags[2][EX_STEPS];
    const int64_t r = blockIdx.x;
    const int32_t maxk = ws.run_maxk[r];
    if (maxk == 0) return;
    const kvm_run run = ws.runs[r];
    const int64_t b = (int64_t)run.start - 1;
    const int64_t total = (int64_t)(maxk - 1) + m;  // steps up to the last point of the last needed window
    const int tid = threadIdx.x;
    const bool consumer = tid < 64;
    const int ptid = tid - 64, pthreads = EX_THREADS - 64;
    const unsigned char *need = ws.need + run.win_base;
    auto produce = [&](int64_t c0, int buf) {
        const int cnt = (int)min((int64_t)EX_STEPS, total - c0);
        ex_produce_replay(x, b, total, m, c0, cnt, bufs + buf * EX_BUF, ptid, pthreads);
        for (int t = ptid; t < cnt; t += pthreads) {
            const int64_t k = c0 + t - (m - 1);  // window completed by step c0 + t
            flags[buf][t] = (k >= 0) ? need[k] : 0;
        }
    };
    double ex = 0.0, ex2 = 0.0;
    const int nch = (int)((total + EX_STEPS - 1) / EX_STEPS);
    if (!consumer) produce(0, 0);
    __syncthreads();
    for (int c = 0; c < nch; c++) {
        const int64_t c0 = (int64_t)c * EX_STEPS;
        if (consumer) {
            if (tid == 0) {
                const double *buf = bufs + (c & 1) * EX_BUF;
                const unsigned char *fl = flags[c & 1];
                const int cnt = (int)min((int64_t)EX_STEPS, total - c0);
                int t = 0;
                const int n_add = (int)max((int64_t)0, min((int64_t)cnt, (int64_t)(m - 1) - c0));  // no window complete yet
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
                    ex += dv.x;   // K/NormQueryEngine.java:498
                    ex2 += dv.y;  // :499
                    if (fl[t]) {
                        const double mu = ex / m;                                    // :507
                        ws.stats[run.win_base + (g - (m - 1))] = make_double2(mu, sqrt(ex2 / m - mu * mu));  // :508
                    }
                    if (g < total - 1) {
                        ex -= ov.x;   // :523
                        ex2 -= ov.y;  // :524
                    }
                }
            }
        } else if (c + 1 < nch) {
            produce(c0 + EX_STEPS, (c + 1) & 1);
        }
        __syncthreads();
    }
}

static hipError_t ex_replay(const double *series, int m, const kvm_replay_ws &ws, hipStream_t s) {
    if (ws.n_runs <= 0) return hipSuccess;
    hipLaunchKernelGGL(kvm_replay_runs_kernel, dim3((unsigned)ws.n_runs), dim3(EX_THREADS), 0, s, series, m, ws);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Exact ED.  q_exact: RSM = the raw query; cNSM = zQ permuted by `order` (|zQ| descending, stable).
template <bool NORM>
__global__ __launch_bounds__(EX_THREADS) void kvm_exact_ed_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                                  const double *__restrict__ q_exact,
                                                                  const int32_t *__restrict__ order,
                                                                  const kvm_surv *__restrict__ surv,
                                                                  const unsigned long long *__restrict__ d_n_surv,
                                                                  kvm_replay_ws ws, int32_t *__restrict__ out_off,
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
        const double *__restrict__ x = series + ((int64_t)sv.start - 1);
        double mean = 0.0, std = 1.0;
        if (NORM) {
            // the reference's mean/std of this window (replayed per run by kvm_replay_runs_kernel)
            const int64_t r = ex_find_run(ws.runs, ws.n_runs, sv.run_start);
            const double2 st = ws.stats[ws.runs[r].win_base + (sv.start - sv.run_start)];
            mean = st.x;
            std = st.y;
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
                               int64_t max_surv, const kvm_replay_ws &ws, int32_t *out_off, double *out_dist,
                               int64_t cap, unsigned long long *counters, int num_cu, hipStream_t s) {
    if (max_surv <= 0) return hipSuccess;
    int64_t blocks = std::min<int64_t>(max_surv, (int64_t)num_cu * 5);  // 32 KiB LDS per workgroup: 5 resident per CU
    if (qd.normalised) {
        hipLaunchKernelGGL((kvm_replay_mark_kernel<kvm_surv>), dim3((unsigned)std::min<int64_t>((max_surv + 255) / 256, 1024)),
                           dim3(256), 0, s, surv, d_n_surv, ws);
        hipError_t e = ex_replay(series, qd.m, ws, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((kvm_exact_ed_kernel<true>), dim3((unsigned)blocks), dim3(EX_THREADS), 0, s, series, qd,
                           q_exact, order, surv, d_n_surv, ws, out_off, out_dist, cap, counters);
    } else {
        hipLaunchKernelGGL((kvm_exact_ed_kernel<false>), dim3((unsigned)blocks), dim3(EX_THREADS), 0, s, series, qd,
                           q_exact, order, surv, d_n_surv, ws, out_off, out_dist, cap, counters);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Exact mean/std for the DTW candidates that reached the DP stage (cNSM-DTW): mark -> replay per run -> apply.  The
// records are updated in place with the reference's mean/std; candidates failing the exact (alpha, beta) test are
// marked with start = -start (the DP kernel skips them).
__global__ __launch_bounds__(256) void kvm_exact_stats_apply_kernel(kvm_query_dev qd, kvm_cand *__restrict__ cands,
                                                                    const unsigned long long *__restrict__ d_n,
                                                                    kvm_replay_ws ws,
                                                                    unsigned long long *__restrict__ counters) {
    const unsigned long long n = *d_n;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        kvm_cand c = cands[i];
        const int64_t r = ex_find_run(ws.runs, ws.n_runs, c.run_start);
        const double2 st = ws.stats[ws.runs[r].win_base + (c.start - c.run_start)];
        c.mean = st.x;
        c.std = st.y;
        if (!kvm_cnsm_filter(st.x, st.y, qd.meanQ, qd.stdQ, qd.alpha, qd.beta)) c.start = -c.start;
        cands[i] = c;
        atomicAdd(&counters[KVM_CNT_RECHECKED], 1ULL);
    }
}

hipError_t kvm_launch_exact_stats(const double *series, const kvm_query_dev &qd, kvm_cand *cands,
                                  const unsigned long long *d_n, int64_t max_n, const kvm_replay_ws &ws,
                                  unsigned long long *counters, int num_cu, hipStream_t s) {
    (void)num_cu;
    if (max_n <= 0) return hipSuccess;
    const unsigned g = (unsigned)std::min<int64_t>((max_n + 255) / 256, 1024);
    hipLaunchKernelGGL((kvm_replay_mark_kernel<kvm_cand>), dim3(g), dim3(256), 0, s, cands, d_n, ws);
    hipError_t e = ex_replay(series, qd.m, ws, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kvm_exact_stats_apply_kernel, dim3(g), dim3(256), 0, s, qd, cands, d_n, ws, counters);
    return hipGetLastError();
}
