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

// lane 0: acc += buf[0..cnt) strictly in order.  Two register sets of 16 values alternate: the LDS reads of the next 16
// are in flight while the current 16 are added, so the chain runs at the latency of a dependent v_add_f64 alone.
__device__ __forceinline__ double ex_seq_add(const double *buf, int cnt, double acc) {
    constexpr int R = 8;  // double2 per set
    double2 a[R], b[R];
    int t = 0;
    if (cnt >= 2 * R) {
#pragma unroll
        for (int u = 0; u < R; u++) a[u] = *reinterpret_cast<const double2 *>(buf + 2 * u);
    }
    while (t + 2 * R <= cnt) {
        const bool more1 = t + 4 * R <= cnt;
        if (more1) {
#pragma unroll
            for (int u = 0; u < R; u++) b[u] = *reinterpret_cast<const double2 *>(buf + t + 2 * R + 2 * u);
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
            acc += a[u].x;
            acc += a[u].y;
        }
        t += 2 * R;
        if (!more1) break;
        const bool more2 = t + 4 * R <= cnt;
        if (more2) {
#pragma unroll
            for (int u = 0; u < R; u++) a[u] = *reinterpret_cast<const double2 *>(buf + t + 2 * R + 2 * u);
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
            acc += b[u].x;
            acc += b[u].y;
        }
        t += 2 * R;
        if (!more2) break;
    }
    for (; t < cnt; t++) acc += buf[t];
    return acc;
}

// Warm-up steps of the running-sum replay (no window complete yet): ex += d, ex2 += d*d for steps [0, n), each step 4
// doubles apart in `buf`; same two-set prefetch.
__device__ __forceinline__ void ex_seq_add2(const double *buf, int n, double &ex, double &ex2) {
    constexpr int R = 8;
    double2 a[R], b[R];
    int t = 0;
    if (n >= R) {
#pragma unroll
        for (int u = 0; u < R; u++) a[u] = *reinterpret_cast<const double2 *>(buf + 4 * u);
    }
    while (t + R <= n) {
        const bool more1 = t + 2 * R <= n;
        if (more1) {
#pragma unroll
            for (int u = 0; u < R; u++) b[u] = *reinterpret_cast<const double2 *>(buf + 4 * (t + R + u));
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
            ex += a[u].x;
            ex2 += a[u].y;
        }
        t += R;
        if (!more1) break;
        const bool more2 = t + 2 * R <= n;
        if (more2) {
#pragma unroll
            for (int u = 0; u < R; u++) a[u] = *reinterpret_cast<const double2 *>(buf + 4 * (t + R + u));
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
            ex += b[u].x;
            ex2 += b[u].y;
        }
        t += R;
        if (!more2) break;
    }
    for (; t < n; t++) {
        ex += buf[4 * t];
        ex2 += buf[4 * t + 1];
    }
}

// Producer side of the running-sum replay: steps [c0, c0+cnt) of the run that starts at 0-based point b; per step
// {d, d*d, o, o*o} where o is the point leaving the window after that step (T[j] of K/NormQueryEngine.java:523).
__device__ __forceinline__ void ex_produce_replay(const double *__restrict__ x, int64_t b, int64_t total, int m, int64_t c0,
                                                  int cnt, double *buf, int ptid, int pthreads) {
    // all loads of this thread's steps are issued before the first is used: one global-load latency per chunk, not
    // one per step (the consumer needs only ~2 us per chunk, so a producer that waits per step would be the bottleneck)
    constexpr int PER = (EX_STEPS + (EX_THREADS - 64) - 1) / (EX_THREADS - 64);
    double d[PER], o[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int t = ptid + i * pthreads;
        const int64_t g = c0 + t;
        d[i] = (t < cnt) ? x[b + g] : 0.0;
        o[i] = (t < cnt && g >= m - 1 && g < total - 1) ? x[b + g - m + 1] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int t = ptid + i * pthreads;
        if (t < cnt) {
            *reinterpret_cast<double2 *>(buf + 4 * t) = make_double2(d[i], d[i] * d[i]);
            *reinterpret_cast<double2 *>(buf + 4 * t + 2) = make_double2(o[i], o[i] * o[i]);
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
        if (atomicMax(&ws.run_maxk[r], k + 1) == 0)  // first mark of this run: queue it for the replay
            ws.run_list[atomicAdd(ws.run_list_n, 1ULL)] = (int32_t)r;
    }
}

__global__ __launch_bounds__(EX_THREADS) void kvm_replay_runs_kernel(const double *__restrict__ x, int m, kvm_replay_ws ws) {
    __shared__ __attribute__((aligned(16))) double bufs[2 * EX_BUF];
    __shared__ unsigned char flags[2][EX_STEPS];
    // one workgroup per QUEUED run (launching one per run of the query costs ~100 us of empty workgroups at 15 000 runs)
    if ((unsigned long long)blockIdx.x >= *ws.run_list_n) return;
  for (unsigned long long li = blockIdx.x; li < *ws.run_list_n; li += gridDim.x) {
    const int64_t r = ws.run_list[li];
    const int32_t maxk = ws.run_maxk[r];
    const kvm_run run = ws.runs[r];
    const int64_t b = (int64_t)run.start - 1;
    const int64_t total = (int64_t)(maxk - 1) + m;  // steps up to the last point of the last needed window
    const int tid = threadIdx.x;
    const bool consumer = tid < 64;
    const int ptid = tid - 64, pthreads = EX_THREADS - 64;
    const unsigned char *need = ws.need + run.win_base;
    auto produce = [&](int64_t c0, int buf) {
        const int cnt = (int)min((int64_t)EX_STEPS, total - c0);
        constexpr int PER = (EX_STEPS + (EX_THREADS - 64) - 1) / (EX_THREADS - 64);
        unsigned char f[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int t = ptid + i * pthreads;
            const int64_t k = c0 + t - (m - 1);  // window completed by step c0 + t
            f[i] = (t < cnt && k >= 0) ? need[k] : 0;
        }
        ex_produce_replay(x, b, total, m, c0, cnt, bufs + buf * EX_BUF, ptid, pthreads);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int t = ptid + i * pthreads;
            if (t < cnt) flags[buf][t] = f[i];
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
                const int n_add = (int)max((int64_t)0, min((int64_t)cnt, (int64_t)(m - 1) - c0));  // no window complete yet
                ex_seq_add2(buf, n_add, ex, ex2);
                // window steps, 4 at a time with the next 4 (values and flags) already in flight
                constexpr int R = 4;
                double2 dv[2][R], ov[2][R];
                unsigned char fv[2][R];
                auto fetch = [&](int t0, int set) {
#pragma unroll
                    for (int u = 0; u < R; u++) {
                        const int tt = min(t0 + u, EX_STEPS - 1);  // (clamped: the tail of a set may lie past cnt, unused)
                        dv[set][u] = *reinterpret_cast<const double2 *>(buf + 4 * tt);
                        ov[set][u] = *reinterpret_cast<const double2 *>(buf + 4 * tt + 2);
                        fv[set][u] = fl[tt];
                    }
                };
                auto step = [&](int t, const double2 d, const double2 o, const unsigned char f) {
                    const int64_t g = c0 + t;
                    ex += d.x;   // K/NormQueryEngine.java:498
                    ex2 += d.y;  // :499
                    // the running sums of a flagged window go out raw: mean = ex/m (:507) and std = sqrt(ex2/m - mean^2)
                    // (:508) are formed by the readers, in parallel -- two divisions and a square root per flagged
                    // window would otherwise sit in this one-lane sequential loop (~400 cycles each)
                    if (f) ws.stats[run.win_base + (g - (m - 1))] = make_double2(ex, ex2);
                    if (g < total - 1) {
                        ex -= o.x;   // :523
                        ex2 -= o.y;  // :524
                    }
                };
                int t = n_add;
                if (t < cnt) fetch(t, 0);
                while (t < cnt) {
                    fetch(t + R, 1);
#pragma unroll
                    for (int u = 0; u < R; u++)
                        if (t + u < cnt) step(t + u, dv[0][u], ov[0][u], fv[0][u]);
                    t += R;
                    if (t >= cnt) break;
                    fetch(t + R, 0);
#pragma unroll
                    for (int u = 0; u < R; u++)
                        if (t + u < cnt) step(t + u, dv[1][u], ov[1][u], fv[1][u]);
                    t += R;
                }
            }
        } else if (c + 1 < nch) {
            produce(c0 + EX_STEPS, (c + 1) & 1);
        }
        __syncthreads();
    }
  }
}

static hipError_t ex_replay(const double *series, int m, const kvm_replay_ws &ws, hipStream_t s) {
    if (ws.n_runs <= 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<int64_t>(ws.n_runs, 2048);
    hipLaunchKernelGGL(kvm_replay_runs_kernel, dim3(grid), dim3(EX_THREADS), 0, s, series, m, ws);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Exact ED.  q_exact: RSM = the raw query; cNSM = zQ permuted by `order` (|zQ| descending, stable).
// A workgroup evaluates EX_GROUP survivors at a time: waves 1-3 produce the distance terms of all of them one chunk
// ahead, lanes 0..EX_GROUP-1 of wave 0 each add one survivor's terms in order (the sequential chains of different
// survivors cost the same instructions as one).
#define EX_GROUP 4
#define EX_GVALS (EX_VALS / EX_GROUP)  // distance terms per survivor per chunk
template <bool NORM>
__global__ __launch_bounds__(EX_THREADS) void kvm_exact_ed_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                                  const double *__restrict__ q_exact,
                                                                  const int32_t *__restrict__ order,
                                                                  const kvm_surv *__restrict__ surv,
                                                                  const unsigned long long *__restrict__ d_n_surv,
                                                                  kvm_replay_ws ws, int32_t *__restrict__ out_off,
                                                                  double *__restrict__ out_dist, int64_t cap,
                                                                  unsigned long long *__restrict__ counters) {
    __shared__ __attribute__((aligned(16))) double bufs[2 * EX_BUF];  // [2][EX_GROUP][EX_GVALS]
    __shared__ double s_mean[EX_GROUP], s_std[EX_GROUP];
    __shared__ int32_t s_start[EX_GROUP], s_active[EX_GROUP];
    __shared__ int32_t s_over;
    const int tid = threadIdx.x;
    const bool consumer = tid < 64;
    const int ptid = tid - 64, pthreads = EX_THREADS - 64;
    const int m = qd.m;
    const double eps2 = qd.eps2;
    const unsigned long long n_surv = *d_n_surv;
    const unsigned long long n_groups = (n_surv + EX_GROUP - 1) / EX_GROUP;
    for (unsigned long long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        if (tid < EX_GROUP) {
            const unsigned long long idx = grp * EX_GROUP + tid;
            int32_t start = 1, active = idx < n_surv;
            double mean = 0.0, std = 1.0;
            if (active) {
                const kvm_surv sv = surv[idx];
                start = sv.start;
                if (NORM) {
                    // the reference's running sums of this window (replayed per run by kvm_replay_runs_kernel)
                    const int64_t r = ex_find_run(ws.runs, ws.n_runs, sv.run_start);
                    const double2 st = ws.stats[ws.runs[r].win_base + (sv.start - sv.run_start)];
                    mean = st.x / m;                     // K/NormQueryEngine.java:507
                    std = sqrt(st.y / m - mean * mean);  // :508
                    if (!kvm_cnsm_filter(mean, std, qd.meanQ, qd.stdQ, qd.alpha, qd.beta)) active = 0;
                    else atomicAdd(&counters[KVM_CNT_PASS_FILTER_EXACT], 1ULL);
                }
            }
            s_start[tid] = start;
            s_active[tid] = active;
            s_mean[tid] = mean;
            s_std[tid] = std;
        }
        if (tid == 0) s_over = 0;
        __syncthreads();
        // distance terms: produced by waves 1-3 one chunk ahead (loads first, arithmetic after: one global-load
        // latency -- two for the gather of cNSM -- per chunk), added in order by the consumer lanes
        auto produce = [&](int c0, double *buf) {
            const int cnt = min(EX_GVALS, m - c0);
            constexpr int PER = (EX_VALS + (EX_THREADS - 64) - 1) / (EX_THREADS - 64);
            double xv[PER], qv[PER];
            int32_t ord[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int e = ptid + i * pthreads, sidx = (e / EX_GVALS) & (EX_GROUP - 1), t = e % EX_GVALS;
                const bool ok = e < EX_VALS && t < cnt && s_active[sidx];
                qv[i] = ok ? q_exact[c0 + t] : 0.0;
                if (NORM) ord[i] = ok ? order[c0 + t] : 0;
                else ord[i] = ok ? c0 + t : 0;
            }
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int e = ptid + i * pthreads, sidx = (e / EX_GVALS) & (EX_GROUP - 1);
                xv[i] = series[(int64_t)s_start[sidx] - 1 + ord[i]];
            }
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int e = ptid + i * pthreads, sidx = (e / EX_GVALS) & (EX_GROUP - 1), t = e % EX_GVALS;
                if (e < EX_VALS && t < cnt && s_active[sidx]) {
                    if (NORM) {
                        const double zv = (xv[i] - s_mean[sidx]) / s_std[sidx];  // K/NormQueryEngine.java:515
                        buf[e] = (zv - qv[i]) * (zv - qv[i]);
                    } else {
                        buf[e] = (xv[i] - qv[i]) * (xv[i] - qv[i]);  // K/QueryEngine.java:357
                    }
                }
            }
        };
        double dist = 0.0;
        const bool mine = consumer && tid < EX_GROUP && s_active[tid & (EX_GROUP - 1)];
        const int nch = (m + EX_GVALS - 1) / EX_GVALS;
        if (!consumer) produce(0, bufs);
        __syncthreads();
        for (int c = 0; c < nch; c++) {
            const int c0 = c * EX_GVALS;
            if (consumer) {
                if (mine) dist = ex_seq_add(bufs + (c & 1) * EX_BUF + tid * EX_GVALS, min(EX_GVALS, m - c0), dist);
                // the reference's loop guard: partial sums only grow, so once every survivor of the group is over
                // the threshold the rest of the window cannot change a decision
                const unsigned long long live = __ballot(mine && dist <= eps2);
                if (tid == 0 && live == 0) s_over = 1;
            } else if (c + 1 < nch) {
                produce(c0 + EX_GVALS, bufs + ((c + 1) & 1) * EX_BUF);
            }
            __syncthreads();
            const bool over = s_over != 0;
            __syncthreads();
            if (over) break;
        }
        if (mine) {
            atomicAdd(&counters[KVM_CNT_RECHECKED], 1ULL);
            if (dist <= eps2) {
                unsigned long long o = atomicAdd(&counters[KVM_CNT_ANSWERS], 1ULL);
                if ((int64_t)o < cap) {
                    out_off[o] = s_start[tid];
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
    // 2-3 workgroups resident per CU (176 / 138 VGPRs), EX_GROUP survivors per workgroup at a time
    int64_t blocks = std::min<int64_t>((max_surv + EX_GROUP - 1) / EX_GROUP, (int64_t)num_cu * (qd.normalised ? 2 : 3));
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
        const double2 st = ws.stats[ws.runs[r].win_base + (c.start - c.run_start)];  // {ex, ex2} of the replay
        const double mean = st.x / qd.m;                        // K/NormQueryEngineDtw.java:550
        const double std = sqrt(st.y / qd.m - mean * mean);     // :551
        c.mean = mean;
        c.std = std;
        if (!kvm_cnsm_filter(mean, std, qd.meanQ, qd.stdQ, qd.alpha, qd.beta)) c.start = -c.start;
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
