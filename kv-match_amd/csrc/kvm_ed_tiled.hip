// kvm_ed_tiled.hip -- LDS-tiled Euclidean verification kernel (RSM-ED, cNSM-ED) for gfx950.
//
// Candidates arrive as runs of consecutive window start offsets, so windows overlap: 32 consecutive
// candidates of length m touch m+31 points, not 32*m.  This kernel stages every point once per tile in LDS and
// reuses it for all candidates of the tile; HBM/L2 traffic drops ~32x below the algorithmic 8*m B per
// candidate and the kernel becomes fp64-VALU-bound (2 VALU ops per (candidate, point) term: v_add_f64 +
// v_fma_f64; 3 for cNSM).
//
// Mapping (one wavefront = 4 independent "rows"; no workgroup barrier anywhere -- every LDS region is
// wave-private, and a wave's LDS operations execute in program order):
//   row   = 16 lanes = one tile of <= 32 consecutive candidates (host splits runs into tiles of <= 32)
//   lane  = 2 adjacent candidates (c0 = 2*lc, c1 = c0+1)
//   All 4 rows of a wave walk j = 0..m-1 in lock-step in chunks of ED_JC points.  Per chunk the wave stages
//   4 x (ED_JC+32) window points and ED_JC query points in LDS; the loads of chunk t+1 are issued into
//   registers before the compute of chunk t and written to LDS after it (the registers are the second
//   buffer).  Inner loop, per 2 points: one ds_read_b128 of the window (X[c0+j+2], X[c0+j+3]) and one
//   broadcast ds_read_b128 of (q[j], q[j+1]) feed 4 terms (c0,j) (c1,j) (c0,j+1) (c1,j+1).
//   Why q is in LDS and not in SGPRs: q (m*8 B = 64 KiB at m=8192) streamed through s_load by 12 waves per CU
//   at different phases thrashes the 16 KiB scalar cache (profiles/r01_ed_notes.md: 41 % of wave cycles in
//   s_waitcnt), and uniform VMEM loads of q are TA-bound.
//   Row stride in LDS is a multiple of 256 B, so the 16-lane groups of ds_read_b128 (which mix lanes of two
//   adjacent rows: {0-3,12-15,20-27}, ...) always cover 16 distinct 16-B slots: conflict-free.
#include "kvm_internal.h"

#define ED_ROWS 4                  // rows per wave
#define ED_ROW_CAND 32             // candidates per row tile
#ifndef ED_JC
#define ED_JC 160                  // points per chunk
#endif
#define ED_ROWLEN (ED_JC + 32)     // staged window points per row per chunk (192 -> 1536 B = 6 * 256 B)
#define ED_LOADS (ED_ROWLEN / 64)  // 64-lane 8-B loads per row per chunk
#define ED_WAVE_LDS ((ED_ROWS + 1) * ED_ROWLEN)  // doubles per wave: 4 rows + q  (7,680 B)
static_assert(ED_ROWLEN % 64 == 0 && (ED_ROWLEN * 8) % 256 == 0, "row stride must keep ds_read_b128 conflict-free");
static_assert(ED_JC % 16 == 0, "chunk = whole groups of 8 points, handled in pairs");

template <bool NORM, bool ABANDON>
__global__ __launch_bounds__(256, 3) void kvm_ed_tiled_kernel(const double *__restrict__ series, int64_t n,
                                                              const double2 *__restrict__ prefix, kvm_query_dev qd,
                                                              const kvm_tile *__restrict__ tiles, int64_t n_tiles,
                                                              kvm_surv *__restrict__ surv, int64_t cap,
                                                              unsigned long long *__restrict__ counters) {
    __shared__ __attribute__((aligned(256))) double kvm_ed_lds[4 * ED_WAVE_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane >> 4, lc = lane & 15;
    const int m = qd.m;
    const double eps2 = qd.eps2;
    const double *__restrict__ q = qd.q;
    double *wlds = kvm_ed_lds + wave * ED_WAVE_LDS;
    double *qlds = wlds + ED_ROWS * ED_ROWLEN;

    // tile descriptors of the wave's four rows (wave-uniform values)
    const int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * ED_ROWS;
    if (t0 >= n_tiles) return;
    int32_t rstart[ED_ROWS], rcount[ED_ROWS], rrun[ED_ROWS];
#pragma unroll
    for (int r = 0; r < ED_ROWS; r++) {
        if (t0 + r < n_tiles) {
            const kvm_tile t = tiles[t0 + r];
            rstart[r] = t.start;
            rcount[r] = t.count;
            rrun[r] = t.run;
        } else {
            rstart[r] = 1;  // empty row: stage from the head of the series (always readable), results discarded
            rcount[r] = 0;
            rrun[r] = 0;
        }
    }
    // static selects (a runtime-indexed register array would be demoted to scratch)
    const int32_t my_start = row == 0 ? rstart[0] : row == 1 ? rstart[1] : row == 2 ? rstart[2] : rstart[3];
    const int32_t my_count = row == 0 ? rcount[0] : row == 1 ? rcount[1] : row == 2 ? rcount[2] : rcount[3];
    const int32_t my_run = row == 0 ? rrun[0] : row == 1 ? rrun[1] : row == 2 ? rrun[2] : rrun[3];
    const int c0 = 2 * lc;
    bool v0 = c0 < my_count, v1 = c0 + 1 < my_count;
    double mean0 = 0.0, sd0 = 1.0, mean1 = 0.0, sd1 = 1.0;
    // pruning thresholds (in units of acc, i.e. before the division by sd^2): eps^2 plus the guard band
    double lim0 = eps2 * (1.0 + 1e-9), lim1 = lim0;
    if (NORM) {
        if (v0) {
            const kvm_stat_guard sg = kvm_window_mean_std_g(prefix, series, (int64_t)my_start - 1 + c0, m, mean0, sd0, kvm_run_hist_of(qd, my_run));
            v0 = kvm_cnsm_filter_guarded(mean0, sd0, qd.meanQ, qd.stdQ, qd.alpha, qd.beta, sg);
            lim0 = (eps2 + kvm_eps2_slack(eps2, m, sg.g, 4.0)) * (sd0 * sd0);
            if (!(sd0 == sd0) || !(sd0 > 0.0) || kvm_guard_degenerate(sg.g)) { mean0 = 0.0; sd0 = 0.0; lim0 = __builtin_huge_val(); }  // exact pass decides
        }
        if (v1) {
            const kvm_stat_guard sg = kvm_window_mean_std_g(prefix, series, (int64_t)my_start + c0, m, mean1, sd1, kvm_run_hist_of(qd, my_run));
            v1 = kvm_cnsm_filter_guarded(mean1, sd1, qd.meanQ, qd.stdQ, qd.alpha, qd.beta, sg);
            lim1 = (eps2 + kvm_eps2_slack(eps2, m, sg.g, 4.0)) * (sd1 * sd1);
            if (!(sd1 == sd1) || !(sd1 > 0.0) || kvm_guard_degenerate(sg.g)) { mean1 = 0.0; sd1 = 0.0; lim1 = __builtin_huge_val(); }
        }
        const unsigned long long b0 = __ballot(v0), b1 = __ballot(v1);
        if (lane == 0 && (b0 | b1))
            atomicAdd(&counters[KVM_CNT_PASS_FILTER], (unsigned long long)(__popcll(b0) + __popcll(b1)));
        if ((b0 | b1) == 0ULL) return;  // nothing in this wave passed the (alpha, beta) filter
        if (!v0) { mean0 = 0.0; sd0 = 1.0; }
        if (!v1) { mean1 = 0.0; sd1 = 1.0; }
    }

    double stage[ED_ROWS + 1][ED_LOADS];
    // Issue the global loads of the chunk starting at point j0 into registers.  Unguarded: the staged series
    // carries KVM_SERIES_PAD zero points past n and the query buffer KVM_SERIES_PAD spare doubles past m.
    auto load_chunk = [&](int j0) {
#pragma unroll
        for (int r = 0; r < ED_ROWS; r++) {
            const double *src = series + ((int64_t)rstart[r] - 1 + j0);
#pragma unroll
            for (int t = 0; t < ED_LOADS; t++) stage[r][t] = src[t * 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < ED_LOADS; t++) stage[ED_ROWS][t] = q[j0 + t * 64 + lane];
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int r = 0; r <= ED_ROWS; r++) {
#pragma unroll
            for (int t = 0; t < ED_LOADS; t++) wlds[r * ED_ROWLEN + t * 64 + lane] = stage[r][t];
        }
    };

    // one (candidate, point) term: RSM (x - q)^2 ; cNSM ((x - mean)/sd - zq)^2 == (x - (mean + zq*sd))^2 / sd^2
    // with the division by sd^2 applied once at the end.
#define ED_TERM(acc, xv, qv, sdv, meanv)                                                        \
    do {                                                                                        \
        const double d_ = NORM ? (xv) - __builtin_fma((qv), (sdv), (meanv)) : (xv) - (qv);      \
        acc = __builtin_fma(d_, d_, acc);                                                       \
    } while (0)
    // 8 points: window values cur, A[0..3] (b128 each), query values Q[0..3] (b128 each)
#define ED_GROUP(A, Q)                                                                          \
    do {                                                                                        \
        ED_TERM(acc0, cur.x, Q[0].x, sd0, mean0);  ED_TERM(acc1, cur.y, Q[0].x, sd1, mean1);    \
        ED_TERM(acc0, cur.y, Q[0].y, sd0, mean0);  ED_TERM(acc1, A[0].x, Q[0].y, sd1, mean1);   \
        ED_TERM(acc0, A[0].x, Q[1].x, sd0, mean0); ED_TERM(acc1, A[0].y, Q[1].x, sd1, mean1);   \
        ED_TERM(acc0, A[0].y, Q[1].y, sd0, mean0); ED_TERM(acc1, A[1].x, Q[1].y, sd1, mean1);   \
        ED_TERM(acc0, A[1].x, Q[2].x, sd0, mean0); ED_TERM(acc1, A[1].y, Q[2].x, sd1, mean1);   \
        ED_TERM(acc0, A[1].y, Q[2].y, sd0, mean0); ED_TERM(acc1, A[2].x, Q[2].y, sd1, mean1);   \
        ED_TERM(acc0, A[2].x, Q[3].x, sd0, mean0); ED_TERM(acc1, A[2].y, Q[3].x, sd1, mean1);   \
        ED_TERM(acc0, A[2].y, Q[3].y, sd0, mean0); ED_TERM(acc1, A[3].x, Q[3].y, sd1, mean1);   \
        cur = A[3];                                                                             \
    } while (0)
#define ED_FETCH(A, Q, g)                                                                       \
    do {                                                                                        \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; t_++) {                                      \
            A[t_] = X2[4 * (g) + 1 + t_];                                                       \
            Q[t_] = Q2[4 * (g) + t_];                                                           \
        }                                                                                       \
    } while (0)

    double acc0 = 0.0, acc1 = 0.0;
    load_chunk(0);
    store_chunk();
    const int nchunks = (m + ED_JC - 1) / ED_JC;
    const double2 *X2 = reinterpret_cast<const double2 *>(wlds + row * ED_ROWLEN + c0);  // X2[i] = (X[c0+2i], X[c0+2i+1])
    const double2 *Q2 = reinterpret_cast<const double2 *>(qlds);
    for (int ch = 0; ch < nchunks; ch++) {
        const int j0 = ch * ED_JC;
        const bool more = ch + 1 < nchunks;
        if (more) load_chunk(j0 + ED_JC);  // global loads in flight during the compute below
        const int jlim = min(ED_JC, m - j0);
        double2 cur = X2[0];
        if (jlim == ED_JC) {
            // full chunk: ED_JC/8 groups of 8 points; the compiler hoists each group's 8 LDS reads to the top of
            // the iteration and 3 waves per SIMD cover their latency
#pragma unroll 2
            for (int g = 0; g < ED_JC / 8; g++) {
                double2 ga[4], qa[4];
                ED_FETCH(ga, qa, g);
                ED_GROUP(ga, qa);
            }
        } else {
            const int jeven = jlim & ~1;
            for (int jj = 0; jj < jeven; jj += 2) {
                const double2 nxt = X2[jj / 2 + 1];
                const double2 qq = Q2[jj / 2];
                ED_TERM(acc0, cur.x, qq.x, sd0, mean0);
                ED_TERM(acc1, cur.y, qq.x, sd1, mean1);
                ED_TERM(acc0, cur.y, qq.y, sd0, mean0);
                ED_TERM(acc1, nxt.x, qq.y, sd1, mean1);
                cur = nxt;
            }
            if (jeven < jlim) {
                const double q0 = qlds[jeven];
                ED_TERM(acc0, cur.x, q0, sd0, mean0);
                ED_TERM(acc1, cur.y, q0, sd1, mean1);
            }
        }
        if (more) {
            if (ABANDON) {
                // partial sums only grow: once both candidates of every lane exceed their pruning threshold the wave is done
                const bool alive = (v0 && !(acc0 > lim0)) || (v1 && !(acc1 > lim1));
                if (!__any(alive)) return;
            }
            store_chunk();  // program order: after this chunk's LDS reads
        }
    }
#undef ED_TERM
#undef ED_GROUP
#undef ED_FETCH
    // survivors of the pruning pass; the exact pass (kvm_exact.hip) computes the reference's distance and decides
    if (v0 && !(acc0 > lim0)) {
        unsigned long long idx = atomicAdd(&counters[KVM_CNT_CANDS1], 1ULL);
        if ((int64_t)idx < cap) surv[idx] = kvm_surv{my_start + c0, my_run};
    }
    if (v1 && !(acc1 > lim1)) {
        unsigned long long idx = atomicAdd(&counters[KVM_CNT_CANDS1], 1ULL);
        if ((int64_t)idx < cap) surv[idx] = kvm_surv{my_start + c0 + 1, my_run};
    }
}

hipError_t kvm_launch_ed_tiled(const double *series, const double2 *prefix, int64_t n, const kvm_query_dev &qd,
                               const kvm_tile *tiles, int64_t n_tiles, kvm_surv *surv, int64_t cap,
                               unsigned long long *counters, hipStream_t s) {
    if (n_tiles <= 0) return hipSuccess;
    const int64_t blocks = (n_tiles + 4 * ED_ROWS - 1) / (4 * ED_ROWS);
    dim3 grid((unsigned)blocks), block(256);
#define LAUNCH(NORM, AB)                                                                                        \
    hipLaunchKernelGGL((kvm_ed_tiled_kernel<NORM, AB>), grid, block, 0, s, series, n, prefix, qd, tiles, n_tiles, \
                       surv, cap, counters)
    if (qd.normalised) {
        if (qd.early_abandon) LAUNCH(true, true); else LAUNCH(true, false);
    } else {
        if (qd.early_abandon) LAUNCH(false, true); else LAUNCH(false, false);
    }
#undef LAUNCH
    return hipGetLastError();
}
