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

#include "kvm_seqadd.h"

#define EX_FENCE2(a, b) asm volatile("" : "+v"(a), "+v"(b)::"memory")

// Producer side of the running-sum replay: steps [c0, c0+cnt) of the run that starts at 0-based point b.  A buffer is
// four arrays of EX_RSTEPS doubles: D[t] = the point entering at step t, D2[t] = D[t]*D[t], O[t] = the point leaving
// the window after that step (T[j] of K/NormQueryEngine.java:523; 0.0 where the reference does not subtract), O2 = O*O.
#define EX_RSTEPS (EX_BUF / 4)  // replay steps per chunk (512)
#define EX_RPROD (EX_THREADS - 128)  // producer threads of the replay kernel (waves 2 and 3)
__device__ __forceinline__ void ex_produce_replay(const double *__restrict__ x, int64_t b, int64_t total, int m, int64_t c0,
                                                  int cnt, double *buf, int ptid) {
    // all loads of this thread's steps are issued before the first is used: one global-load latency per chunk, not
    // one per step; unconditional loads from clamped (always valid) addresses, selection afterwards
    constexpr int PER = (EX_RSTEPS + EX_RPROD - 1) / EX_RPROD;
    double d[PER], o[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int t = ptid + i * EX_RPROD;
        const int64_t g = c0 + t;
        const int64_t gd = g < total ? g : total - 1, go = g >= m - 1 ? g - m + 1 : 0;
        const double dv = x[b + gd], ov = x[b + (go < total ? go : total - 1)];
        d[i] = (t < cnt) ? dv : 0.0;
        o[i] = (t < cnt && g >= m - 1 && g < total - 1) ? ov : 0.0;
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int t = ptid + i * EX_RPROD;
        if (t < EX_RSTEPS) {  // (steps past cnt are written as zeros: the consumers' prefetch may read them)
            buf[t] = d[i];
            buf[EX_RSTEPS + t] = d[i] * d[i];
            buf[2 * EX_RSTEPS + t] = o[i];
            buf[3 * EX_RSTEPS + t] = o[i] * o[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Running-sum replay, one workgroup per run.  Survivors of one run share the prefix of the reference's running sums,
// so the run is replayed ONCE, up to the last window any survivor needs, and {mean, std} is stored for every flagged
// window.  (Replaying per survivor would repeat the prefix for each of them.)
template <typename REC>
__global__ __launch_bounds__(256) void kvm_replay_mark_kernel(const REC *__restrict__ recs,
                                                              const unsigned long long *__restrict__ d_n,
                                                              kvm_replay_ws ws) {
    const unsigned long long n = *d_n;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const REC rec = recs[i];
        if (rec.start < 0) continue;
        const int64_t r = rec.run;
        const int32_t k = rec.start - ws.runs[r].start;
        if (atomicMax(&ws.run_maxk[r], k + 1) == 0)  // first mark of this run: queue it for the replay
            ws.run_list[atomicAdd(ws.run_list_n, 1ULL)] = (int32_t)r;
    }
}

#ifdef KVM_EX_TIMING  // diagnostics only (profiles/micro/replay_timing.hip): cycle stamps of block 0's consumer and one producer
__device__ unsigned long long kvm_dbg[8192];
#define EX_STAMP(slot) do { if (blockIdx.x == 0) kvm_dbg[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define EX_STAMP(slot) do { } while (0)
#endif
__global__ __launch_bounds__(EX_THREADS) void kvm_replay_runs_kernel(const double *__restrict__ x, int m, kvm_replay_ws ws) {
    // Two consumer waves: lane 0 of wave 0 runs the `ex` chain, lane 0 of wave 1 the `ex2` chain (they never meet: the
    // readers combine them).  A lone wavefront issues one fp64 operation per ~10 cycles whether or not it depends on
    // the previous one (profiles/micro/chain_bench.hip: two interleaved chains take twice as long as one), so two
    // chains in one lane cost 20 cycles per step; on two waves they cost 10.  Waves 2-3 produce.
    __shared__ __attribute__((aligned(16))) double bufs[2 * EX_BUF];
    // one workgroup per QUEUED run (launching one per run of the query costs ~100 us of empty workgroups at 15 000 runs)
    if ((unsigned long long)blockIdx.x >= *ws.run_list_n) return;
  for (unsigned long long li = blockIdx.x; li < *ws.run_list_n; li += gridDim.x) {
    const int64_t r = ws.run_list[li];
    const int32_t maxk = ws.run_maxk[r];
    const kvm_run run = ws.runs[r];
    const int64_t b = (int64_t)run.start - 1;
    const int64_t total = (int64_t)(maxk - 1) + m;  // steps up to the last point of the last needed window
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const bool consumer = wave < 2;
    const int ptid = tid - 128;
    auto produce = [&](int64_t c0, int buf) {
        const int cnt = (int)min((int64_t)EX_RSTEPS, total - c0);
        ex_produce_replay(x, b, total, m, c0, cnt, bufs + buf * EX_BUF, ptid);
    };
    double acc = 0.0;  // ex (wave 0) or ex2 (wave 1)
    const int nch = (int)((total + EX_RSTEPS - 1) / EX_RSTEPS);
    if (!consumer) produce(0, 0);
    __syncthreads();
    for (int c = 0; c < nch; c++) {
        const int64_t c0 = (int64_t)c * EX_RSTEPS;
        if (consumer) {
            if ((tid & 63) == 0) {
                if (tid == 0) EX_STAMP(4 * c + 0);
                // wave 0 reads D / O, wave 1 reads D2 / O2
                const double *dbuf = bufs + (c & 1) * EX_BUF + wave * EX_RSTEPS, *obuf = dbuf + 2 * EX_RSTEPS;
                const int cnt = (int)min((int64_t)EX_RSTEPS, total - c0);
                const int n_add = (int)max((int64_t)0, min((int64_t)cnt, (int64_t)(m - 1) - c0));  // no window complete yet
                acc = ex_seq_add(dbuf, n_add, acc);  // K/NormQueryEngine.java:498 / :499
                // Window steps.  Branch-free: the departing point's term is 0.0 where the reference does not subtract
                // (x - 0.0 == x bit for bit), and the running sum is stored for EVERY window up to the last one a
                // survivor needs (the readers look only at theirs), so a round is straight-line code whose LDS reads
                // stay one round ahead.  ws.stats plane 0 = ex, plane 1 = ex2: each consumer stores its own plane.
                {
                    constexpr int R = 4;  // steps per register set
                    double dv[2][R], ov[2][R];
                    double a_l = acc, dummy = 0.0;
                    double *st_out = ws.stats + (int64_t)wave * ws.stats_plane + run.win_base + (c0 - (int64_t)(m - 1));
                    const int nw = cnt - n_add, rounds = nw / (2 * R);
                    auto fetch = [&](int t0, int set) {
#pragma unroll
                        for (int u = 0; u < R; u++) {
                            dv[set][u] = dbuf[t0 + u];
                            ov[set][u] = obuf[t0 + u];
                        }
                    };
                    auto consume = [&](int t0, int set) {
#pragma unroll
                        for (int u = 0; u < R; u++) {
                            a_l += dv[set][u];              // K/NormQueryEngine.java:498 / :499
                            st_out[t0 + u] = a_l;           // -> mean, std (:507-508) in the readers
                            a_l -= ov[set][u];              // :523 / :524
                        }
                    };
                    if (rounds > 0) fetch(n_add, 0);
                    for (int rr = 0; rr < rounds; rr++) {
                        const int t = n_add + rr * 2 * R;
                        fetch(t + R, 1);
                        EX_FENCE2(a_l, dummy);
                        consume(t, 0);
                        fetch((rr + 1 < rounds) ? t + 2 * R : n_add, 0);
                        EX_FENCE2(a_l, dummy);
                        consume(t + R, 1);
                    }
                    for (int t = n_add + rounds * 2 * R; t < cnt; t++) {
                        a_l += dbuf[t];
                        st_out[t] = a_l;
                        a_l -= obuf[t];
                    }
                    acc = a_l;
                }
                if (tid == 0) EX_STAMP(4 * c + 1);
            }
        } else if (c + 1 < nch) {
            if (tid == 128) EX_STAMP(4 * c + 2);
            produce(c0 + EX_RSTEPS, (c + 1) & 1);
            if (tid == 128) EX_STAMP(4 * c + 3);
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

hipError_t kvm_launch_replay_runs(const double *series, int m, const kvm_replay_ws &ws, hipStream_t s) {
    return ex_replay(series, m, ws, s);
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
                    const kvm_run rn = ws.runs[sv.run];
                    const int64_t wi = rn.win_base + (sv.start - rn.start);
                    mean = ws.stats[wi] / m;                                     // K/NormQueryEngine.java:507
                    std = sqrt(ws.stats[ws.stats_plane + wi] / m - mean * mean);  // :508
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
                const int kk = c0 + t < m ? c0 + t : m - 1;  // clamped: loads are unconditional (a load under a
                const double qq = q_exact[kk];                // condition becomes a branch with its own wait)
                qv[i] = ok ? qq : 0.0;
                if (NORM) {
                    const int32_t oo = order[kk];
                    ord[i] = ok ? oo : 0;
                } else {
                    ord[i] = ok ? kk : 0;
                }
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
                               int64_t cap, unsigned long long *counters, int num_cu, hipStream_t s, const kvm_surv *ties,
                               const unsigned long long *d_n_ties, int64_t max_ties) {
    if (max_surv <= 0) return hipSuccess;
    // 2-3 workgroups resident per CU (176 / 138 VGPRs), EX_GROUP survivors per workgroup at a time
    int64_t blocks = std::min<int64_t>((max_surv + EX_GROUP - 1) / EX_GROUP, (int64_t)num_cu * (qd.normalised ? 2 : 3));
    if (qd.normalised) {
        hipLaunchKernelGGL((kvm_replay_mark_kernel<kvm_surv>), dim3((unsigned)std::min<int64_t>((max_surv + 255) / 256, 1024)),
                           dim3(256), 0, s, surv, d_n_surv, ws);
        if (ties)  // (their runs join the same replay)
            hipLaunchKernelGGL((kvm_replay_mark_kernel<kvm_surv>), dim3((unsigned)std::min<int64_t>((max_ties + 255) / 256, 1024)),
                               dim3(256), 0, s, ties, d_n_ties, ws);
        hipError_t e = ex_replay(series, qd.m, ws, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((kvm_exact_ed_kernel<true>), dim3((unsigned)blocks), dim3(EX_THREADS), 0, s, series, qd,
                           q_exact, order, surv, d_n_surv, ws, out_off, out_dist, cap, counters);
        if (ties) {
            e = kvm_launch_ucr_ed_ties(series, qd, q_exact, order, ties, d_n_ties, max_ties, ws, out_off, out_dist, cap, counters,
                                       num_cu, s);
            if (e != hipSuccess) return e;
        }
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
        const kvm_run rn = ws.runs[c.run];
        const int64_t wi = rn.win_base + (c.start - rn.start);  // ex, ex2 of the replay
        const double mean = ws.stats[wi] / qd.m;                                      // K/NormQueryEngineDtw.java:550
        const double std = sqrt(ws.stats[ws.stats_plane + wi] / qd.m - mean * mean);  // :551
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
