// kvm_ed.hip -- Euclidean verification kernels (RSM-ED, cNSM-ED) for gfx950.
//
// Replaces the inner loops of K/QueryEngine.java:354-362 and K/NormQueryEngine.java:497-521.
// fp64 throughout.  kvm_ed_wave_kernel: one wavefront per candidate window, coalesced 8 B/lane loads of the window
// straight from HBM/L2, per-lane partial sums, 64-lane butterfly reduction.  Bound: HBM (isolated candidates, 8*m B each)
// or L2 (overlapping candidates of one interval).  Not the default (kvm_ed_dot.hip is): kept as the independent
// cross-check the tests run every ED case through (KVM_OPT_ED_KERNEL = 1).
// The decision quantity is dist^2 <= eps^2 exactly as the reference writes it (epsilon*epsilon).
#include "kvm_internal.h"

#define ED_TILE_MAX 64      // candidates per tile (host splits runs into tiles of at most this many)
#define ED_CHECK_EVERY 2048 // early-abandon test granularity (points)

template <bool NORM, bool ABANDON>
__global__ __launch_bounds__(256) void kvm_ed_wave_kernel(const double *__restrict__ series,
                                                          const double2 *__restrict__ prefix, kvm_query_dev qd,
                                                          const kvm_tile *__restrict__ tiles,
                                                          kvm_surv *__restrict__ surv, int64_t cap,
                                                          unsigned long long *__restrict__ counters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const kvm_tile tile = tiles[blockIdx.x];
    const int m = qd.m;
    const double eps2 = qd.eps2;
    const double *__restrict__ q = qd.q;
    unsigned passed = 0;

    for (int c = wave; c < tile.count; c += 4) {
        const int32_t start = tile.start + c;                  // 1-based window start
        const double *__restrict__ x = series + (start - 1);  // window
        double mean = 0.0, inv = 1.0, lim = eps2 * (1.0 + 1e-9);  // pruning threshold = eps^2 + guard band
        if (NORM) {
            double std;
            const kvm_stat_guard sg = kvm_window_mean_std_g(prefix, series, (int64_t)start - 1, m, mean, std, kvm_run_hist_of(qd, tile.run));
            if (!kvm_cnsm_filter_guarded(mean, std, qd.meanQ, qd.stdQ, qd.alpha, qd.beta, sg)) continue;  // wave-uniform
            passed++;
            inv = 1.0 / std;
            lim = eps2 + kvm_eps2_slack(eps2, m, sg.g, 4.0);
            if (!(std == std) || !(std > 0.0) || kvm_guard_degenerate(sg.g)) { inv = 0.0; lim = __builtin_huge_val(); }  // exact pass decides
        }
        double acc = 0.0;
        bool dead = false;
        for (int j0 = 0; j0 < m; j0 += ED_CHECK_EVERY) {
            const int jend = min(j0 + ED_CHECK_EVERY, m);
#pragma unroll 8
            for (int j = j0 + lane; j < jend; j += 64) {
                double v = x[j];
                if (NORM) v = (v - mean) * inv;
                const double d = v - q[j];
                acc = __builtin_fma(d, d, acc);
            }
            if (ABANDON && jend < m) {
                // partial sums only grow: once the wave total exceeds eps^2 the candidate is not an answer
                if (kvm_wave_sum(acc) > lim) {
                    dead = true;
                    break;
                }
            }
        }
        if (dead) continue;
        const double total = kvm_wave_sum(acc);
        if (!(total > lim) && lane == 0) {  // (NaN total: a NaN / Inf point in the window -- the exact pass rejects it)
            unsigned long long idx = atomicAdd(&counters[KVM_CNT_CANDS1], 1ULL);
            if ((int64_t)idx < cap) surv[idx] = kvm_surv{start, tile.run};
        }
    }
    if (NORM && lane == 0 && passed) atomicAdd(&counters[KVM_CNT_PASS_FILTER], (unsigned long long)passed);
}

hipError_t kvm_launch_ed_wave(const double *series, const double2 *prefix, int64_t n, const kvm_query_dev &qd,
                              const kvm_tile *tiles, int64_t n_tiles, kvm_surv *surv, int64_t cap,
                              unsigned long long *counters, hipStream_t s) {
    (void)n;
    if (n_tiles <= 0) return hipSuccess;
    dim3 grid((unsigned)n_tiles), block(256);
#define LAUNCH(NORM, AB)                                                                                       \
    hipLaunchKernelGGL((kvm_ed_wave_kernel<NORM, AB>), grid, block, 0, s, series, prefix, qd, tiles, surv, cap, \
                       counters)
    if (qd.normalised) {
        if (qd.early_abandon) LAUNCH(true, true); else LAUNCH(true, false);
    } else {
        if (qd.early_abandon) LAUNCH(false, true); else LAUNCH(false, false);
    }
#undef LAUNCH
    return hipGetLastError();
}
