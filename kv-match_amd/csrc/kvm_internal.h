// kvm_internal.h -- shared declarations of libkvmatch_hip (gfx950 only; no CPU fallback anywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/kvmatch.h"

#define KVM_WAVE 64

// Block-restarted, mean-shifted prefix sums.  For block b (KVM_PS_BLOCK points) with shift c_b = x[first point
// of b]:  P[i] = { sum (x - c_b), sum (x - c_b)^2 } from the start of i's block up to i inclusive.
// A window's sums are differences of O(block)-sized partial sums of SHIFTED values, so neither the n-sized
// magnitude of a global prefix nor the E[x^2] - mean^2 cancellation of large-offset data (random walks reach
// |x| ~ 1e3) enters: var = S2/m - (S1/m)^2 is formed around c_b ~ window level.  Mathematically identical to the
// reference's mean = ex/m, std = sqrt(ex2/m - mean*mean) (K/NormQueryEngine.java:507-508), but more accurate
// than the reference's own running sums.  See DESIGN.md "mu/sigma".
#define KVM_PS_BLOCK_LOG2 14
#define KVM_PS_BLOCK (1 << KVM_PS_BLOCK_LOG2)
// The staged series is padded with this many zero points so that tile staging may read past n unguarded.
#define KVM_SERIES_PAD 2048

// One candidate run = a clipped interval of consecutive window start offsets (1-based).
struct kvm_run {
    int32_t start;      // first window start offset (1-based)
    int32_t count;      // number of consecutive offsets
    int64_t tile_base;  // index of the run's first tile in the tile list (class A list in hybrid mode)
    int64_t tile_base_b;  // hybrid ED tiling only: index of the run's class-B tile (the first <= 32 of its remainder), or -1
    int64_t tile_base_c;  // ... of its class-C tile (the last <= 16 of its remainder), or -1
    int64_t win_base;   // ordinal of the run's first window among all windows of the query (exclusive prefix of counts)
};

// Scratch of the running-sum replay (kvm_exact.hip): one {ex, ex2} slot per window of the query, one "windows
// needed" count per run, the queue of runs to replay.
struct kvm_replay_ws {
    const kvm_run *runs;
    int64_t n_runs;
    int32_t *run_maxk;     // [n_runs]  number of leading windows of the run that must be replayed (0 = none)
    double *stats;         // [2][windows] plane 0 = ex, plane 1 = ex2: the reference's running sums when the window completes
    int64_t stats_plane;   // windows (offset of plane 1); one plane per consumer wave: contiguous 8-byte stores
    int32_t *run_list;     // [n_runs]  indices of the runs with run_maxk > 0, in marking order
    unsigned long long *run_list_n;  // number of entries (counters[KVM_CNT_REPLAY_RUNS])
};

// Tile = up to `count` consecutive candidates taken from one run.
struct kvm_tile {
    int32_t start;      // first window start offset of the tile (1-based)
    int32_t count;      // consecutive offsets in the tile
    int32_t run_start;  // first window start offset of the tile's whole run (where the reference's running sums restart)
    int32_t run;        // index of that run in the query's run list (lists with repeated `left` values hold several runs
                        // with the same start: the start alone does not identify the run)
};

// Survivor of a fast ED kernel, to be decided by the exact pass (kvm_exact.hip).
struct kvm_surv {
    int32_t start;
    int32_t run;  // index of its run in the run list
};

// Survivor record flowing between the cascade kernels.
struct kvm_cand {
    int32_t start;      // window start offset (1-based); negated once the exact (alpha, beta) test rejects it
    int32_t run;        // index of its run in the run list
    double mean;
    double std;
    double lim;         // cNSM: pruning limit of the lower bounds for THIS candidate (eps^2 + its guard band, stage 1)
};

struct kvm_dev_buf {
    void *p = nullptr;
    size_t cap = 0;
};

struct kvm_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_staged = nullptr;  // recorded after the last H2D copy out of h_pinned (guards its reuse)
    bool staged_pending = false;
    hipEvent_t ev_done = nullptr;    // recorded at the very end of a kvm_verify_device call, behind everything that reads the
    hipStream_t done_stream = nullptr;  // context's device scratch: a later call on ANOTHER stream makes that stream wait for
    bool done_pending = false;       // it (hipStreamWaitEvent: a device-side wait, the host goes on preparing the call)
    int64_t n = 0;
    double *d_series = nullptr;
    bool own_series = false;
    double2 *d_prefix = nullptr;  // n entries {sum x, sum x^2}, block-restarted
    double *d_blockmax = nullptr; // max x^2 of every prefix block (NaN if the block holds one): kvm_run_hist
    // cached data envelope for LB_Keogh2 (per rho)
    double *d_env_u = nullptr, *d_env_l = nullptr;
    int32_t env_rho = -1;
    hipEvent_t ev_env = nullptr;     // recorded behind the envelope build
    hipStream_t env_stream = nullptr;  // the stream it was built on
    // grow-only scratch
    kvm_dev_buf b_query, b_tiles, b_out_off, b_out_dist, b_counters, b_cands, b_cands2, b_aux, b_aux2, b_scratch, b_plans, b_ucr, b_edx;
    void *h_pinned = nullptr;
    void *h_readback2 = nullptr;  // pinned: answers beyond the first KVM_SPEC_ANSWERS (grow-only)
    size_t h_readback2_cap = 0;
    void *h_readback = nullptr;  // pinned, device-mapped: the readback kernel stores counters + first answers here
    size_t h_pinned_cap = 0;
    int64_t opt_early_abandon = 1, opt_lb_cascade = 1, opt_ed_kernel = 0, opt_timing = 0, opt_dp_kernel = 0,
            opt_sort_violation = 0, opt_ucr_loops = 0, opt_strict_bounds = 1, opt_in_flight = 1, opt_ed_force = 0,
            opt_serialize_enqueue = 0, last_ed_shape = 0;
    int run_in_flight = 0;   // contexts of the kvm_run_jobs / kvm_run_device_jobs call this one is part of right now (kvm_set_run_in_flight)
    int num_cu = 256;
    int64_t last_kernel_ns = 0;
    bool timing_pending = false;  // ev0/ev1 recorded, elapsed time not read yet (read lazily: no mid-pipeline sync)
    std::string err;
};

struct kvm_result {
    std::vector<int32_t> off;
    std::vector<double> dist;
    int64_t candidates = 0;
    double elapsed_ms = 0;
    int64_t stats[KVM_STAT_COUNT] = {0};
};

// ---- kernel launchers (each file implements its own) -----------------------------------------------
// common
hipError_t kvm_launch_bswap64(uint64_t *d, int64_t n, hipStream_t s);
// (both expansion kernels also zero the KVM_CNT_TOTAL counters and, when `run_maxk` is given, the replay scratch)
// One interval of the candidate list -> its clipped run and tile counts (window start offsets of [left, right]:
// max(1, left-shift) .. min(right-shift, n-m+1), K/QueryEngine.java:346-354).  The host sums these over the list
// (build_plan: launch geometry, buffer sizes), the device turns them into the runs with their prefix sums
// (kvm_plan_runs_kernel); both use this one rule.
struct kvm_plan_item {
    int64_t lo;        // first window start
    int64_t c64;       // hi - lo + 1 before clamping at zero (the RSM range check needs it: c64 + m - 1 points are fetched)
    uint32_t c;        // windows (0: clipped away)
    uint32_t tiles_a;  // tiles (class A tiles in hybrid mode)
    uint32_t tiles_b;  // hybrid mode: 1 if the run has a class-B remainder tile
    uint32_t tiles_c;  // hybrid mode: 1 if the run has a class-C remainder tile
};
// Tile classes of the correlation-form ED pruner (kvm_ed_dot.hip): class A = up to 64 consecutive candidates (one row of
// 8 lanes x 8 candidates), class B = up to 32 (8 lanes x 4), class C = up to 16 (half a row: 4 lanes x 4; a row holds two
// unrelated class-C tiles).  What is left of a run after its full 64-candidate tiles (rem = count mod 64) becomes
//     rem >= KVM_EDA_MIN (33): one more class-A tile      17 .. 32: one class-B tile      1 .. 16: one class-C tile
// (measured, profiles/ed_class_cost.py: per lane-slot classes A and B cost the same, class C 1.2 x as much because its
// chunks are half as long -- a win below 17 candidates only; a partly filled class-A tile runs faster than a full one,
// so cutting 33..48 into a class-B and a class-C tile buys nothing, which KVM_EDA_MIN = 49 would do).  The device-built
// lists of the retile kernels use classes A and B only (they are latency-bound: see retile_collect).
#define KVM_PLAN_SUMS_BYTES 40  // sizeof(plan_sums) of kvm_common.hip: per-block totals of kvm_launch_plan_runs
#define KVM_EDB_TILE 32u
#define KVM_EDC_TILE 16u
#define KVM_EDA_MIN 33u
#ifdef __HIPCC__
__host__ __device__
#endif
static inline void kvm_ed_remainder_classes(uint32_t rem, uint32_t &a, uint32_t &b, uint32_t &c) {
    a = rem >= KVM_EDA_MIN ? 1u : 0u;
    b = (rem > KVM_EDC_TILE && rem < KVM_EDA_MIN) ? 1u : 0u;
    c = (rem >= 1u && (rem <= KVM_EDC_TILE || (rem > KVM_EDB_TILE && rem < KVM_EDA_MIN))) ? 1u : 0u;
}
#ifdef __HIPCC__
__host__ __device__
#endif
// `hybrid`: 0 = uniform tiles of 2^tile_log2, 1 = classes A / B (a remainder of <= 16 goes to class B too), 2 = A / B / C
static inline void kvm_plan_tiles(uint32_t c, int hybrid, int tile_log2, uint32_t &tiles_a, uint32_t &tiles_b, uint32_t &tiles_c) {
    if (hybrid) {
        uint32_t a1;
        kvm_ed_remainder_classes(c & 63u, a1, tiles_b, tiles_c);
        tiles_a = (c >> 6) + a1;
        if (hybrid == 1) {
            tiles_b += tiles_c;
            tiles_c = 0u;
        }
    } else {
        tiles_a = (c + (1u << tile_log2) - 1u) >> tile_log2;
        tiles_b = tiles_c = 0u;
    }
}
#ifdef __HIPCC__
__host__ __device__
#endif
static inline kvm_plan_item kvm_plan_interval(int64_t l, int64_t r, int64_t shift, int64_t hi_max, int hybrid, int tile_log2) {
    kvm_plan_item it;
    int64_t lo = l - shift, hi = r - shift;
    lo = lo < 1 ? 1 : lo;
    hi = hi > hi_max ? hi_max : hi;
    const int64_t c64 = hi - lo + 1;
    it.lo = lo;
    it.c64 = c64;
    it.c = c64 > 0 ? (uint32_t)c64 : 0u;
    kvm_plan_tiles(it.c, hybrid, tile_log2, it.tiles_a, it.tiles_b, it.tiles_c);
    return it;
}
// `chunk_lo` / `chunk_hi` (optional, per interval; may be pinned host memory): the data range [lo, hi] (1-based points) of
// the chunk the reference fetches for that interval (cNSM engines: greedy-merged chunks, K/NormQueryEngineDtw.java:492-520);
// copied per non-empty run into `run_chunk` -- where the reference's data envelope is clipped (DTW only).
hipError_t kvm_launch_plan_runs(const int32_t *src_left, const int32_t *src_right, int32_t *left, int32_t *right, int32_t k,
                                int32_t shift, int64_t hi_max, int hybrid, int tile_log2, kvm_run *runs, void *scratch,
                                hipStream_t s, const int32_t *chunk_lo = nullptr, const int32_t *chunk_hi = nullptr,
                                int2 *run_chunk = nullptr);
hipError_t kvm_launch_upload(void *dst, const void *src_pinned, size_t bytes, hipStream_t s);
hipError_t kvm_launch_set_u64(unsigned long long *a, unsigned long long va, unsigned long long *b, unsigned long long vb,
                              hipStream_t s);
// (run_hist != nullptr, cNSM: also the per-run maximum of the block maxima `blockmax` over the run's data range)
// (`up_*`: optionally the call's query block, copied from the pinned staging buffer by the same launch)
hipError_t kvm_launch_expand_tiles(const kvm_run *runs, int64_t n_runs, int32_t tile_max, kvm_tile *tiles,
                                   unsigned long long *counters, int32_t *run_maxk, hipStream_t s,
                                   const double *blockmax = nullptr, double *run_hist = nullptr, int32_t m = 0, int64_t n = 0,
                                   void *up_dst = nullptr, const void *up_src_pinned = nullptr, size_t up_bytes = 0);
// hybrid ED tiling: classes A / B / C of kvm_ed_remainder_classes
hipError_t kvm_launch_expand_tiles_hybrid(const kvm_run *runs, int64_t n_runs, int hybrid, kvm_tile *tilesA, kvm_tile *tilesB,
                                          kvm_tile *tilesC, unsigned long long *counters, int32_t *run_maxk, hipStream_t s,
                                          void *up_dst = nullptr, const void *up_src_pinned = nullptr, size_t up_bytes = 0);
hipError_t kvm_launch_prefix(const double *x, double2 *prefix, int64_t n, hipStream_t s);
// max x^2 per prefix block (at load)
hipError_t kvm_launch_blockmax(const double *x, int64_t n, double *blockmax, hipStream_t s);
// counters + the first `spec` answers, stored straight into device-mapped pinned host memory:
// [KVM_CNT_TOTAL x u64 | spec x int32 | spec x f64]
#define KVM_SPEC_ANSWERS 4096
hipError_t kvm_launch_readback(const unsigned long long *counters, const int32_t *off, const double *dist, int64_t spec,
                               void *host, hipStream_t s);
hipError_t kvm_launch_copy_u64(unsigned long long *dst, const unsigned long long *src, hipStream_t s);  // *dst = src ? *src : 0
hipError_t kvm_launch_pack_answers(const int32_t *off, const double *dist, const int64_t *count, int64_t pack_cap,
                                   double2 *packed, hipStream_t s);
hipError_t kvm_launch_window_means(const double *x, const double2 *prefix, int32_t w, int64_t first0, int64_t count,
                                   double *out, hipStream_t s);
hipError_t kvm_launch_envelope(const double *x, int64_t n, int32_t rho, double *u, double *l, hipStream_t s);

// Query-side constants shared by kernels.
struct kvm_query_dev {
    const double *q;   // raw query (RSM) or z-normalised query (cNSM), device, m entries
    const double *qu;  // upper envelope of q (DTW), device
    const double *ql;  // lower envelope of q (DTW), device
    int32_t m;
    int32_t rho;
    double eps2;    // epsilon * epsilon
    double meanQ, stdQ, alpha, beta;
    int32_t normalised;  // 0 RSM, 1 cNSM
    int32_t early_abandon;
    // correlation-form pruner (kvm_ed_dot.hip): statistics of the device query q[] (raw for RSM, zQ for cNSM)
    double q_mean;          // mean of q
    const double *qc;       // q - q_mean, device, m entries (+ KVM_SERIES_PAD spare)
    const double2 *q_pref8; // q_pref8[t] = { sum_{j < t*ED8_JC} (q_j - q_mean), sum_{j < t*ED8_JC} (q_j - q_mean)^2 }, t = 0 .. ceil(m/ED8_JC)
                            // (entries past m hold the totals)
    const double *qk;       // the phases of qc for the 16-per-lane class A (kvm_ed_dot.hip, EDD16_KARATSUBA): three arrays of
                            // kvm_qk_len(m) doubles -- qc[2t] | qc[2t + 1] | qc[2t] + qc[2t + 1] -- zero beyond the query
    const double *run_hist; // cNSM: per run, the largest x^2 among the points the reference's running sums of that run pass
                            // over (to the granularity of the prefix blocks; NaN with a NaN among them) -- see kvm_hist_*
};

// ED pruners: survivors (decision quantity within the guard band of eps^2, filter within its guard band) are appended
// to `surv` (count in counters[KVM_CNT_CANDS1]); the exact pass decides.  Wave-per-candidate form (the cross-check).
hipError_t kvm_launch_ed_wave(const double *series, const double2 *prefix, int64_t n, const kvm_query_dev &qd,
                              const kvm_tile *tiles, int64_t n_tiles, kvm_surv *surv, int64_t cap,
                              unsigned long long *counters, hipStream_t s);
#ifdef __HIPCC__
__host__ __device__
#endif
static inline int kvm_qk_len(int m) { return (m + 1) / 2 + KVM_SERIES_PAD / 2; }
#define ED4_JC 160  // chunk lengths of the correlation-form pruner (also the granularity of kvm_query_dev::q_pref / q_pref8 /
#define ED8_JC 128  // q_pref16): class B (4 candidates per lane), class A (8 per lane), class C (4 per lane, two 16-candidate
#define ED16_JC 80  // tiles per row)
// host: classes A / B (1) or A / B / C (2) for a host-planned list (the launcher's model of the launch)
int kvm_ed_dot_choose_classes(int m, int64_t tilesA, int64_t tilesB, int64_t tilesC, int num_cu, int early_abandon);
size_t kvm_ed_dot_x_bytes();          // the pruner's cross-workgroup scratch: sums first, then its arrival counters (kvm_ed_dot_x_counters_at)
size_t kvm_ed_dot_x_counters_at();
hipError_t kvm_launch_ed_dot(const double *series, const double2 *prefix, int64_t n, const kvm_query_dev &qd,
                             const kvm_tile *tilesA, int64_t nA, const kvm_tile *tilesB, int64_t nB,
                             const kvm_tile *tilesC, int64_t nC, kvm_surv *surv,
                             int64_t cap, unsigned long long *counters, int num_cu, const unsigned long long *d_counts,
                             int32_t *out_off, double *out_dist, int64_t out_cap, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                             int in_flight = 1, int64_t force = 0, int64_t *shape = nullptr,    // in_flight: launches of other contexts
                             double *xpart = nullptr, unsigned *xcnt = nullptr);   // scratch of kvm_ed_dot_x_bytes() bytes (xcnt zero): tiny device-built lists
                                                       // expected next to this one (kvm_in_flight); force: KVM_OPT_ED_FORCE; shape: KVM_OPT_ED_LAST_SHAPE
static inline int kvm_in_flight(const kvm_ctx *ctx) { return (int)std::max<int64_t>(std::max<int64_t>(ctx->opt_in_flight, ctx->run_in_flight), 1); }
// cNSM: 64-candidate tiles -> trimmed class-A / class-B tile lists of the candidates passing the guarded (alpha, beta)
// filter; the list lengths go to d_counts[0..1] (kvm_launch_ed_dot then takes nA, nB as upper bounds; no class C here)
hipError_t kvm_launch_retile_rsm(const double *series, const kvm_query_dev &qd, const kvm_tile *tiles, int64_t n_tiles,
                                 kvm_tile *tilesA, kvm_tile *tilesB, unsigned long long *d_counts, hipStream_t s);
hipError_t kvm_launch_retile_cnsm(const double *series, const double2 *prefix, const kvm_query_dev &qd, const kvm_tile *tiles,
                                  int64_t n_tiles, kvm_tile *tilesA, kvm_tile *tilesB, unsigned long long *d_counts,
                                  hipStream_t s);
// Exact final pass (kvm_exact.hip)
// (`ties`, cNSM with KVM_OPT_UCR_LOOPS: the windows kvm_launch_ucr_ed_screen handed to the executor's literal loop -- replayed
// and evaluated behind the survivors, kvm_ucr.hip)
hipError_t kvm_launch_exact_ed(const double *series, const kvm_query_dev &qd, const double *q_exact,
                               const int32_t *order, const kvm_surv *surv, const unsigned long long *d_n_surv,
                               int64_t max_surv, const kvm_replay_ws &ws, int32_t *out_off, double *out_dist,
                               int64_t cap, unsigned long long *counters, int num_cu, hipStream_t s,
                               const kvm_surv *ties = nullptr, const unsigned long long *d_n_ties = nullptr, int64_t max_ties = 0);
// kvm_ucr.hip: the executors' loop semantics (KVM_OPT_UCR_LOOPS)
hipError_t kvm_launch_ucr_ed_screen(const double *series, const double2 *prefix, const kvm_query_dev &qd, const double *zqs,
                                    const int32_t *order, const kvm_tile *tilesA, int64_t nA, const kvm_tile *tilesB, int64_t nB,
                                    const unsigned long long *d_counts, kvm_surv *ties, int64_t cap, unsigned long long *counters,
                                    int num_cu, hipStream_t s);
// replay of the runs queued in ws.run_list (kvm_exact.hip)
hipError_t kvm_launch_replay_runs(const double *series, int m, const kvm_replay_ws &ws, hipStream_t s);
hipError_t kvm_launch_ucr_ed_ties(const double *series, const kvm_query_dev &qd, const double *zqs, const int32_t *order,
                                  const kvm_surv *ties, const unsigned long long *d_n, int64_t max_ties, const kvm_replay_ws &ws,
                                  int32_t *out_off, double *out_dist, int64_t cap, unsigned long long *counters, int num_cu,
                                  hipStream_t s);
hipError_t kvm_launch_exact_stats(const double *series, const kvm_query_dev &qd, kvm_cand *cands,
                                  const unsigned long long *d_n, int64_t max_n, const kvm_replay_ws &ws,
                                  unsigned long long *counters, int num_cu, hipStream_t s);

// DTW cascade.
//
// The reference's answer set is NOT {candidates with banded DTW <= eps^2}: DtwUtils.dtw returns `min_cost + cb[i+r+1]` as
// soon as that lower bound is >= bsf (K/utils/DtwUtils.java:324-326), the LB loops stop at the first partial sum >= bsf
// (:210, :242) and every caller accepts `value <= eps^2` -- so a partial sum that lands EXACTLY on eps^2 passes, with
// partially filled cb arrays behind it, and a row bound that lands exactly on eps^2 is reported as an answer at distance
// eps.  On continuous data such ties have measure zero, on integer-valued data they are common.  The cascade here
// reproduces them:
//   * screens (approximate mean/std, any arithmetic) may only drop a candidate when the reference's own sequence of
//     partial sums, IN THE REFERENCE'S ORDER, crosses eps^2 cleanly: the last partial below eps^2 - slack, the next above
//     eps^2 + slack (kvm_retile_dtw_kernel for LB_Kim, kvm_dtw_screen_kernel for LB_Keogh);
//   * the DP kernel decides everything else with the replayed (exact) mean/std: LB_Kim as the reference computes it, both
//     LB_Keogh loops as wave-parallel ordered scans of exactly rounded terms (a prefix within tau of eps^2 is ambiguous),
//     the row minima of the DP (tracked alongside the anti-diagonal sweep) against the suffix sums of the chosen bound;
//   * what is ambiguous -- some compared quantity within rounding of eps^2 -- goes to the sequential, reference-ordered
//     path (kvm_dtw_ref.hip) and comes back with exact cb[] arrays for a DP run that compares without tolerance.
struct kvm_dtw_aux {
    const int32_t *order;        // [m] summation order of the reference's LB_Keogh loops (identity for RSM-DTW)
    const double *uo, *lo, *qo;  // [m] query envelope and query in that order (K/NormQueryEngineDtw.java:480-490)
    const double *env_u, *env_l; // [n] data envelope of radius rho over the whole series (interior of every chunk)
    const kvm_run *runs;         // runs of the query (chunk bounds: RSM = the run's own data range)
    int64_t n_runs;
    const int2 *run_chunk;       // cNSM: [n_runs] data range of the run's chunk; nullptr for RSM
    double *patch;               // scratch: 8 * rho doubles per DP wave (running max / min at the chunk ends)
    kvm_cand *ambig;             // out: candidates whose decision needs the sequential path
    const double *cb_exact;      // in (second DP run only): [2m per candidate] exact cb arrays, see kvm_dtw_ref.hip
    double tau;                  // relative tolerance between a wave-parallel sum (terms formed with a reciprocal) and
                                 // the reference's sequential one
};
hipError_t kvm_launch_dtw_screen(const double *series, const double2 *prefix, const kvm_query_dev &qd, const kvm_dtw_aux &ax,
                                 const kvm_tile *tiles, int64_t n_tiles, const unsigned long long *d_n_tiles, kvm_cand *out,
                                 unsigned long long *counters, hipStream_t s);
// sequential reference-ordered evaluation of the ambiguous candidates: LB_Kim, both LB_Keogh loops with their exits,
// the choice of the larger bound and its suffix sums (K/NormQueryEngineDtw.java:555-587); rejected candidates get
// start = -start, the others their cb[] in slots[2m * index + (lim != 0 ? m : 0)]
// (keep_slots: the cb1 / cb2 slots already hold what the arrays hold when the loops start -- KVM_OPT_UCR_LOOPS -- instead of
// being zeroed like the engines' fresh arrays)
hipError_t kvm_launch_dtw_ref_chains(const double *series, const kvm_query_dev &qd, const kvm_dtw_aux &ax, kvm_cand *cands,
                                     int64_t n, double *slots, hipStream_t s, bool keep_slots = false);
// cNSM-DTW with KVM_OPT_UCR_LOOPS: slots[2m per candidate] <- what the executor's never-cleared cb1 / cb2 hold when the
// candidate's loops start (kvm_launch_dtw_ref_chains then runs with keep_slots).  depth1/2: [windows], tmax1/2: [n_tiles]
// scratch, filled on the first call of a query (`tables_ready` = false).
hipError_t kvm_launch_ucr_dtw_stale(const double *series, const kvm_query_dev &qd, const kvm_dtw_aux &ax, const kvm_tile *tiles,
                                    int64_t n_tiles, const kvm_replay_ws &ws, int32_t *depth1, int32_t *depth2, int32_t *tmax1,
                                    int32_t *tmax2, const kvm_cand *cands, int64_t n, double *slots, bool tables_ready,
                                    hipStream_t s);
hipError_t kvm_launch_dtw_filter(const double *series, const double2 *prefix, int64_t n, const kvm_query_dev &qd,
                                 const kvm_tile *tiles, int64_t n_tiles, int32_t use_lb, kvm_cand *out,
                                 unsigned long long *counters, hipStream_t s);
// DTW: filter + LB_Kim for every candidate (one lane each), 64-candidate tiles -> trimmed <= 32-candidate tiles of the
// survivors; list length -> counters[KVM_CNT_TILES_A], PASS_FILTER / PASS_KIM counted here
hipError_t kvm_launch_retile_dtw(const double *series, const double2 *prefix, const kvm_query_dev &qd, const kvm_tile *tiles,
                                 int64_t n_tiles, kvm_tile *out_tiles, unsigned long long *counters, hipStream_t s);
// waves of the DP launch that use a slice of kvm_dtw_aux::patch (the launch never starts more)
int64_t kvm_dtw_dp_max_waves(int num_cu);
size_t kvm_dtw_plan_bytes(int64_t n);  // scratch of kvm_launch_dtw_dp: one plan record per candidate + the compact list of
                                       // the candidates that reach the DP
hipError_t kvm_launch_dtw_dp(const double *series, int64_t n, const kvm_query_dev &qd, const kvm_dtw_aux &ax, const kvm_cand *in,
                             const unsigned long long *d_n_in, int64_t max_in, void *plan_buf, int32_t *out_off, double *out_dist,
                             int64_t cap, unsigned long long *counters, int num_cu, int dp_form, hipStream_t s,
                             hipEvent_t ev_dp_start = nullptr);  // recorded between the prologue kernel and the DP
// dp_form = KVM_OPT_DP_KERNEL: 0 auto, 1 generic LDS form, 2 one candidate per wave in registers, 3 packed register form
// kvm_dtw_pair.hip: the packed register form (two candidates per wave, any V); kvm_dtw_pack_v: its cells per lane, 0 = none
int kvm_dtw_pack_v(int rho);
hipError_t kvm_launch_dtw_dp_pack(const double *series, const kvm_query_dev &qd, const kvm_dtw_aux &ax, const kvm_cand *in,
                                  const void *plans, const unsigned *list, int share_mode, int64_t max_in, int32_t *out_off,
                                  double *out_dist, int64_t cap, unsigned long long *counters, int num_cu, hipStream_t s);

// kvm_selftest.hip: what the device under the run sustains (kvm_measure_device)
hipError_t kvm_measure_device_rate(int what, int num_cu, hipStream_t s, hipEvent_t e0, hipEvent_t e1, double *value);

// counters layout (unsigned long long[KVM_CNT_TOTAL])
#define KVM_CNT_ANSWERS 0
#define KVM_CNT_WINDOWS 1
#define KVM_CNT_PASS_FILTER 2
#define KVM_CNT_PASS_KIM 3
#define KVM_CNT_PASS_KEOGH 4
#define KVM_CNT_PASS_KEOGH2 5
#define KVM_CNT_QUEUE 6   // work-queue head of the DP kernel
#define KVM_CNT_CANDS1 7  // #records in cands (after filter/Kim)
#define KVM_CNT_CANDS2 8  // #records in cands2 (after Keogh)
#define KVM_CNT_RECHECKED 9           // evaluations done by the exact pass
#define KVM_CNT_PASS_FILTER_EXACT 10  // candidates passing the exact (alpha, beta) test
#define KVM_CNT_REPLAY_RUNS 11        // runs queued for the running-sum replay
#define KVM_CNT_TILES_A 12            // cNSM-ED: trimmed tile lists built on the device (kvm_retile_cnsm_kernel)
#define KVM_CNT_TILES_B 13
#define KVM_CNT_TILES_C 14
#define KVM_CNT_AMBIG 15              // DTW: candidates handed to the sequential reference-ordered path
#define KVM_CNT_UCR_TIES 16           // KVM_OPT_UCR_LOOPS, cNSM-ED: windows handed to the executor's literal distance() loop
#define KVM_CNT_DP_RERUN 17           // DTW: candidates swept a second time with their row minima in full (see dp_reg_sweep)
#define KVM_CNT_DP_LIST 18            // DTW: candidates the prologue passes on to the DP (entries of the compact list)
#define KVM_CNT_QUEUE2 19             // DTW: work-queue head of the one-candidate-per-wave DP kernel behind the packed one
#define KVM_CNT_TOTAL 20

// kvm_index.hip: index build step 1 on the device
struct kvm_index_dev {
    unsigned long long *keys = nullptr;  // key bit patterns, one per interval
    int32_t *first = nullptr, *last = nullptr;
    int64_t total = 0;
    // after kvm_index_group_device: intervals ordered by (key descending, position), one group per distinct key
    unsigned long long *group_keys = nullptr;
    int32_t *group_off = nullptr;  // [n_groups + 1]
    int64_t n_groups = 0;
    void *blocks[16] = {};
};
int kvm_index_intervals_device(kvm_ctx *ctx, int32_t w, int64_t count, kvm_index_dev *d, hipStream_t s);
int kvm_index_group_device(kvm_index_dev *d, hipStream_t s);
void kvm_index_dev_free(kvm_index_dev *d);


// ---- device helpers ---------------------------------------------------------------------------------
#ifdef __HIPCC__
// Shifted sums of the 0-based range [s, s+len): returns S1 = sum(x - c), S2 = sum(x - c)^2 and the shift c (= the
// first point of the block that contains s), from the block-restarted shifted prefix sums.
// `pm` (optional): the largest magnitude behind the subtraction -- max of the prefix entries P.y used and of the
// re-basing terms k*dc^2 -- from which kvm_prefix_err bounds the absolute rounding error of s1 and s2: a spike earlier
// in the block leaves the prefix sums large long after it has left every window.
__device__ __forceinline__ void kvm_window_shifted_sums(const double2 *__restrict__ P, const double *__restrict__ x,
                                                        int64_t s, int32_t len, double &s1, double &s2, double &c,
                                                        double *pm = nullptr) {
    double mag = 0.0;
    const int64_t e = s + len - 1;
    const int64_t bs = s >> KVM_PS_BLOCK_LOG2, be = e >> KVM_PS_BLOCK_LOG2;
    c = x[bs << KVM_PS_BLOCK_LOG2];  // shift of the first block; everything is re-based to it
    const bool at_start = (s & (KVM_PS_BLOCK - 1)) == 0;
    if (bs == be) {
        const double2 pe = P[e];
        mag = fabs(pe.y);  // (P.y is non-decreasing inside a block: P[s-1].y <= P[e].y)
        if (at_start) {
            s1 = pe.x;
            s2 = pe.y;
        } else {
            const double2 ps = P[s - 1];
            s1 = pe.x - ps.x;
            s2 = pe.y - ps.y;
        }
    } else {
        const double2 pl = P[((bs + 1) << KVM_PS_BLOCK_LOG2) - 1];  // tail of the first block
        mag = fabs(pl.y);
        if (at_start) {
            s1 = pl.x;
            s2 = pl.y;
        } else {
            const double2 ps = P[s - 1];
            s1 = pl.x - ps.x;
            s2 = pl.y - ps.y;
        }
        for (int64_t b = bs + 1; b <= be; b++) {  // following blocks (exactly one unless len > KVM_PS_BLOCK)
            const int64_t first = b << KVM_PS_BLOCK_LOG2;
            const int64_t lastp = b < be ? first + KVM_PS_BLOCK - 1 : e;
            const double2 t = P[lastp];
            const double dc = x[first] - c;  // re-base from c_b to c: sum(x-c) = T1 + k*dc, sum(x-c)^2 = T2 + 2dc*T1 + k*dc^2
            const double k = (double)(lastp - first + 1);
            s1 += t.x + k * dc;
            s2 += t.y + (2.0 * dc) * t.x + k * (dc * dc);
            mag = fmax(mag, fabs(t.y) + k * (dc * dc));
        }
    }
    if (pm) *pm = mag;
}

// Absolute error bounds of (s1, s2) as returned by kvm_window_shifted_sums, from the magnitude `pm` behind them.
// kvm_prefix_kernel adds <= ~28 roundings per entry: |dP.y| <= 3e-15 P.y, and by Cauchy-Schwarz over <= 16 384 points
// |dP.x| <= 28 u sum|x - c_b| <= 4e-13 sqrt(P.y); two to three entries are combined.  Margins of ~50.  NaN in, NaN out
// (every pruning test is written so that NaN keeps the candidate).
struct kvm_sum_err {
    double e1, e2;
};
// The reference's running sums (ex += d, ex2 += d * d, later -= the same) keep the rounding residue of everything that
// ever passed through them since the interval began.  While values of magnitude X fill the window, ex2 is about m X^2 and
// each of the ~2m additions and subtractions that carry them in and out rounds by up to u m X^2: up to 2 m^2 u X^2 stays
// in ex2 -- 2 m u X^2 in the variance ex2 / m -- for the rest of the run, whatever the windows look like by then; the
// mean keeps up to 2 m u X.  With `hist` = the largest x^2 the run has seen (u = 1.1e-16, margin 2):
__device__ __forceinline__ double kvm_hist_var(double hist, int32_t m) { return 4e-16 * m * hist; }
__device__ __forceinline__ double kvm_hist_mean(double hist, int32_t m) { return 4e-16 * m * sqrt(hist); }
// On ordinary data (hist ~ level^2) this is of the size of the level term of kvm_sigma_guard; after a burst 1e6 x the
// local sigma it decides (found by the adversarial sweep, seed 23: a 1e-158-scaled series with O(1) planted windows).
__device__ __forceinline__ kvm_sum_err kvm_prefix_err(double pm) {
    kvm_sum_err r;
    r.e1 = 2e-11 * sqrt(pm);
    r.e2 = 1e-12 * pm + 1e-314;  // + the absolute quantum of squares that land among the denormals (|x| ~ 1e-155 and below):
                                 //   <= 2.5e-324 per product, up to ~4e9 products behind a running or a prefix sum
    return r;
}

// Two-phase form of kvm_window_shifted_sums: all loads first (unconditional, clamped addresses), arithmetic later --
// lets a kernel put the loads of several windows in flight together instead of paying 2-3 dependent latencies each.
// raw operands of kvm_window_shifted_sums for a window spanning at most two prefix blocks (loads only, so that they
// can be issued a chunk ahead of their use)
struct kvm_raw_sums {
    double2 pe, ps, pl;
    double c0, c1;
    bool at_start;
    int32_t kind;  // 0: one block, 1: two blocks, 2: more (recomputed by the generic routine)
    double k1;     // points of the window inside its last block
};

__device__ __forceinline__ kvm_raw_sums kvm_issue_window_sums(const double2 *__restrict__ P, const double *__restrict__ x, int64_t s,
                                                 int32_t len) {
    // branch-free: every load is issued unconditionally from an address that is valid in all cases
    kvm_raw_sums r;
    const int64_t e = s + len - 1;
    const int64_t bs = s >> KVM_PS_BLOCK_LOG2, be = e >> KVM_PS_BLOCK_LOG2;
    const int64_t first_e = be << KVM_PS_BLOCK_LOG2, tail_s = ((bs + 1) << KVM_PS_BLOCK_LOG2) - 1;
    r.c0 = x[bs << KVM_PS_BLOCK_LOG2];
    r.c1 = x[first_e];
    r.pe = P[e];
    r.pl = P[tail_s < e ? tail_s : e];
    r.ps = P[s > 0 ? s - 1 : 0];
    r.at_start = (s & (KVM_PS_BLOCK - 1)) == 0;  // then the window starts its block and nothing is subtracted
    r.kind = bs == be ? 0 : (be == bs + 1 ? 1 : 2);
    r.k1 = (double)(e - first_e + 1);
    return r;
}
// same arithmetic, in the same order, as kvm_window_shifted_sums
__device__ __forceinline__ void kvm_finish_window_sums(const kvm_raw_sums &r, const double2 *__restrict__ P, const double *__restrict__ x,
                                                int64_t s, int32_t len, double &s1, double &s2, double &c,
                                                double *pm = nullptr) {
    if (r.kind == 2) {
        kvm_window_shifted_sums(P, x, s, len, s1, s2, c, pm);
        return;
    }
    c = r.c0;
    const double psx = r.at_start ? 0.0 : r.ps.x, psy = r.at_start ? 0.0 : r.ps.y;
    if (r.kind == 0) {
        s1 = r.pe.x - psx;
        s2 = r.pe.y - psy;
    } else {
        s1 = r.pl.x - psx;
        s2 = r.pl.y - psy;
        const double dc = r.c1 - c;
        s1 += r.pe.x + r.k1 * dc;
        s2 += r.pe.y + (2.0 * dc) * r.pe.x + r.k1 * (dc * dc);
    }
    if (pm) *pm = r.kind == 0 ? fabs(r.pe.y) : fmax(fabs(r.pl.y), fabs(r.pe.y) + r.k1 * ((r.c1 - c) * (r.c1 - c)));
}


// mean / std of the window 0-based [s, s+m) from the shifted block prefix sums.
__device__ __forceinline__ void kvm_window_mean_std(const double2 *__restrict__ P, const double *__restrict__ x,
                                                    int64_t s, int32_t m, double &mean, double &std) {
    double s1, s2, c;
    kvm_window_shifted_sums(P, x, s, m, s1, s2, c);
    const double ms = s1 / m;  // mean - c
    mean = c + ms;
    std = sqrt(s2 / m - ms * ms);
}

// (alpha, beta) filter, K/NormQueryEngine.java:511.
__device__ __forceinline__ bool kvm_cnsm_filter(double mean, double std, double meanQ, double stdQ, double alpha,
                                                double beta) {
    return fabs(mean - meanQ) <= beta && (std / stdQ) <= alpha && (std / stdQ) >= 1.0 / alpha;
}

// ---- guard bands of the pruning kernels (the exact pass decides inside them) ----
// Every pruning test has the form `drop only if value > limit`, so a NaN anywhere (a NaN / Inf point poisons the prefix
// sums up to the end of its block) keeps the candidate for the exact pass, which follows the reference's arithmetic.
//
// g = relative uncertainty between this library's sigma and the reference's (running sums, sqrt(ex2/m - mean^2)):
//   * the reference loses about E[x^2]/var ulps times the drift of ~1e4 additions;
//   * `gx`: this library's own share, from the prefix magnitudes (kvm_prefix_err): d(var)/(2 var) + d(mean)/sigma.
// g >= KVM_G_DEGENERATE (also NaN / zero variance): mean and sigma carry no information, nothing may be pruned on them.
#define KVM_G_DEGENERATE 0.25
__device__ __forceinline__ double kvm_sigma_guard(double mean, double std, double gx = 0.0) {
    const double var = std * std;
    const double g = 1e-9 + 1e-12 * ((mean * mean + var) / var) + gx;
    return (g < KVM_G_DEGENERATE) ? g : KVM_G_DEGENERATE;  // also catches NaN / zero variance
}
__device__ __forceinline__ bool kvm_guard_degenerate(double g) { return !(g < KVM_G_DEGENERATE); }
// What the guarded filter needs besides mean and sigma: the guard, the absolute uncertainty of the mean, and an upper
// bound on the sigma the REFERENCE can arrive at (its own cancellation noise included) for windows whose variance is
// lost in rounding -- a constant stretch has sigma 0 here and 1e-7-ish noise (or NaN) there.
struct kvm_stat_guard {
    double g, mean_err, std_hi;
};
__device__ __forceinline__ kvm_stat_guard kvm_make_stat_guard(double c, double s1, double s2, int32_t m, double pm,
                                                              double hist = 0.0) {
    kvm_stat_guard r;
    const double ms = s1 / m, var = s2 / m - ms * ms, mean = c + ms;
    const kvm_sum_err e = kvm_prefix_err(pm);
    r.mean_err = e.e1 / m + kvm_hist_mean(hist, m);
    const double var_err = e.e2 / m + 2.0 * fabs(ms) * r.mean_err + kvm_hist_var(hist, m);
    r.g = kvm_sigma_guard(mean, sqrt(var), var_err / (2.0 * var) + r.mean_err / sqrt(var));
    r.std_hi = sqrt(fmax(var, 0.0) + var_err + 1e-8 * (mean * mean + fabs(var)));
    return r;
}
// mean / std / guard in one go (prefix-magnitude share included)
__device__ __forceinline__ kvm_stat_guard kvm_window_mean_std_g(const double2 *__restrict__ P, const double *__restrict__ x,
                                                                int64_t s, int32_t m, double &mean, double &std,
                                                                double hist = 0.0) {
    double s1, s2, c, pm;
    kvm_window_shifted_sums(P, x, s, m, s1, s2, c, &pm);
    const double ms = s1 / m;
    mean = c + ms;
    std = sqrt(s2 / m - ms * ms);
    return kvm_make_stat_guard(c, s1, s2, m, pm, hist);
}
// `hist` of a tile's run (0 for the RSM engines: no running sums there)
__device__ __forceinline__ double kvm_run_hist_of(const kvm_query_dev &qd, int32_t run) { return qd.run_hist ? qd.run_hist[run] : 0.0; }
// (alpha, beta) test widened by the guard.  With a degenerate guard (NaN sigma, variance rounded below zero, prefix sums
// too coarse for this window) only the one thing that is still certain is tested: a sigma that cannot reach the lower
// limit; everything else is left to the exact pass.
__device__ __forceinline__ bool kvm_cnsm_filter_guarded(double mean, double std, double meanQ, double stdQ, double alpha,
                                                        double beta, const kvm_stat_guard &sg) {
    if (!(std == std) || kvm_guard_degenerate(sg.g)) return !(sg.std_hi / stdQ < (1.0 / alpha) * 0.5);
    const double gm = 1e-9 * fmax(1.0, fmax(fabs(mean), fabs(meanQ))) + sg.mean_err;
    const double ratio = std / stdQ;
    return !(fabs(mean - meanQ) > beta + gm) && !(ratio > alpha * (1.0 + sg.g)) && !(ratio < (1.0 / alpha) * (1.0 - sg.g));
}
// Slack on eps^2 for a z-normalised distance (or lower bound) computed with a mean / sigma that are off by relative g:
// first order 2 g sqrt(m dist) at dist ~ eps^2, second order g^2 m (what is left when eps = 0).  Infinite for a
// degenerate guard.
__device__ __forceinline__ double kvm_eps2_slack(double eps2, int32_t m, double g, double factor) {
    if (kvm_guard_degenerate(g)) return __builtin_huge_val();
    return eps2 * 1e-9 + factor * (g * sqrt((double)m * eps2) + g * g * m);
}


// Wave-uniform values as SCALARS.  A value every lane agrees on must also look uniform to the compiler, or it lowers the
// branches that depend on it as divergent control flow -- and a loop with a cross-lane operation (ds_bpermute, DPP) at
// its head and "divergent" continue edges can then be restructured so that the operation runs with some lanes masked
// off (seen in the DP kernels' work-queue loop: lane 0, which pops the queue, was parked in an outer loop while the other
// lanes spun on a stale index).  v_readfirstlane / v_readlane put the value into SGPRs.
__device__ __forceinline__ unsigned long long kvm_uniform_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double kvm_readlane_f64(double v, int src_lane) {  // src_lane must be wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double kvm_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double kvm_wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
#endif
