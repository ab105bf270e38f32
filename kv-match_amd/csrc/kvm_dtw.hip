// kvm_dtw.hip -- the deciding stage of the DTW cascade (RSM-DTW, cNSM-DTW) for gfx950.
//
// Replaces K/QueryEngineDtw.java:401-451 and K/NormQueryEngineDtw.java:540-599 plus the static functions of
// K/utils/DtwUtils.java, for the candidates the screens (kvm_dtw_lb.hip) could not certainly drop.  One wavefront per
// candidate, with the replayed (reference-exact) mean/std:
//   prologue : lbKimHierarchy as written (:149-189); lbKeoghCumulative and lbKeoghDataCumulative (:206-257) as
//              wave-parallel ORDERED scans of exactly rounded terms -- the reference stops each loop at its first partial
//              sum >= eps^2 and passes on iff the value is <= eps^2; the data envelope clipped at the ends of the fetched
//              chunk as lowerUpperLemire builds it (:50-91);
//   DP       : Sakoe-Chiba banded DP swept by anti-diagonals; every cell performs exactly the reference's operations,
//              min(min(x,y),z) + (a-b)*(a-b) (:316), in fp64 without contraction, so the cell values are bit-identical to
//              the reference's regardless of the order in which cells are visited;
//   row test : the reference returns `min_cost + cb[i+r+1]` at the first row where that is >= eps^2 (:324-326) and the
//              callers accept `<= eps^2`: the row minima are tracked alongside the sweep and tested against the suffix
//              sums of the larger LB_Keogh bound's terms (K/NormQueryEngineDtw.java:576-587).
// Whatever lands within rounding of eps^2 in a wave-parallel sum is "ambiguous" and goes to the sequential path
// (kvm_dtw_ref.hip), which returns exact cb[] arrays for a second run of the same DP that compares without tolerance.
// With KVM_OPT_LB_CASCADE=0 the screens are skipped and this stage sees every filter-passing candidate: same answers.
#include "kvm_dtw_dp.h"


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
            const kvm_stat_guard sg = kvm_window_mean_std_g(prefix, series, (int64_t)start - 1, m, mean, std, kvm_run_hist_of(qd, tile.run));
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
        c.run = tile.run;
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


#ifndef DP_SUM_MODE
#define DP_SUM_MODE 1
#endif
// The cascade in front of the DP for one candidate (K/NormQueryEngineDtw.java:555-587), wave-cooperative.
__device__ int dp_prologue(const double *__restrict__ series, const kvm_query_dev &qd, const kvm_dtw_aux &ax,
                           const kvm_cand &c, int lane, double *__restrict__ patch, dp_plan &pl) {
    const int m = qd.m, r = qd.rho;
    const double E = qd.eps2, mean = c.mean, sd = c.std;
    const double *__restrict__ x = series + (c.start - 1);
#ifdef DP_STAMPS
    const unsigned long long ts0 = wall_clock64();
    unsigned long long tsv[4] = {0, 0, 0, 0};
    int tsn = 0;
#define DP_STAMP(tag)                                                                                    \
    do {                                                                                                 \
        if (tsn < 4) tsv[tsn++] = wall_clock64() - ts0;                                                  \
        if (tsn == 4 && lane == 0 && blockIdx.x == 0 && (threadIdx.x >> 6) == 0) {                       \
            printf("stamps kim+chunk %llu patch %llu round %llu totals %llu\n", tsv[0], tsv[1], tsv[2], tsv[3]); \
            tsn = 5;                                                                                     \
        }                                                                                                \
    } while (0)
#else
#define DP_STAMP(tag)
#endif
    if (m >= 6) {  // (below that the reference's bound indexes outside the window; every entry point requires m >= 25)
        const double kim = dref_kim(x, qd.q, m, mean, sd, E);
        if (!(kim <= E)) return DP_REJECT;  // :556
    }
    const double Elo = E * (1.0 - ax.tau), Ehi = E * (1.0 + ax.tau);
    // eps^2 == 0: the loops `lb < bsf` do not even start and row 0 of the DP already "reaches" bsf -- sequential path
    if (!(0.0 < Elo)) return DP_AMBIG;
    // ---- the data envelope at the ends of the fetched chunk ----
    dref_chunk(ax, m, c.run, pl.cbeg, pl.cend);
    pl.sb = max(pl.cbeg, pl.cend - 2 * r + 1);
    DP_STAMP("kim+chunk");
    dp_build_patch(series, r, pl, c.start, m, lane, patch);
    DP_STAMP("patch");
    // ---- lbKeoghCumulative and lbKeoghDataCumulative: terms in the reference's order, inclusive scans = its partial sums
    // up to rounding.  The two loops are independent of each other (each fills its own cb array from scratch), so they are
    // scanned in the same rounds -- four 64-term chunks of each per round, loads and divisions issued together: a round is
    // otherwise one exposed global-load latency and a surviving candidate needs m / 64 of them per loop.  The reference
    // runs lbK first and lbK2 only if lbK passes; a CLEAN crossing of either (last partial certainly below eps^2, next
    // certainly above) drops the candidate whatever the other does: if lbK stops on an exact tie it passes and lbK2,
    // evaluated in full, drops it; if lbK2 would never be reached, lbK has dropped it already.  A crossing within rounding
    // of eps^2 makes the candidate ambiguous unless the other loop drops it cleanly.
    double carry1 = 0.0, carry2 = 0.0;
    bool amb1 = false, amb2 = false;  // the loop has stopped at a partial within rounding of eps^2
    const double inv_sd = 1.0 / sd;
    // (the positions of a round's terms are fetched one round ahead: a round is then ONE exposed memory latency -- the gathers
    // they address -- instead of two)
    int onext[4];
#pragma unroll
    for (int h = 0; h < 4; h++) onext[h] = ax.order[min(64 * h + lane, m - 1)];
    for (int base = 0; base < m && !(amb1 && amb2); base += 256) {
        double d1[4], d2[4];
        int ocur[4];
#pragma unroll
        for (int h = 0; h < 4; h++) {
            ocur[h] = onext[h];
            onext[h] = ax.order[min(base + 256 + 64 * h + lane, m - 1)];
        }
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const int i = base + 64 * h + lane, ic = min(i, m - 1);
            const int o = ocur[h];
            const double t1 = dp_term_q(x[o], mean, inv_sd, ax.uo[ic], ax.lo[ic]);
            double U, L;
            dp_env_at(ax, patch, r, pl, c.start + o, U, L);
            const double t2 = dp_term_d(ax.qo[ic], U, L, mean, inv_sd);
            d1[h] = i < m ? t1 : 0.0;
            d2[h] = i < m ? t2 : 0.0;
        }
        // Past the first round (a candidate whose totals reach eps^2: one of its loops stops somewhere): a round whose own
        // total leaves both running sums below eps^2 (1 - tau) holds no stop -- the partial sums never decrease -- and is
        // passed over with two wave reductions instead of eight ordered scans; only the round of the stop is scanned.
        if (base > 0) {
            double r1 = (d1[0] + d1[1]) + (d1[2] + d1[3]), r2 = (d2[0] + d2[1]) + (d2[2] + d2[3]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {   // (both reductions in the same butterfly steps: their shuffles overlap)
                const double p1 = __shfl_xor(r1, o, 64), p2 = __shfl_xor(r2, o, 64);
                r1 += p1;
                r2 += p2;
            }
            if ((amb1 || carry1 + r1 < Elo) && (amb2 || carry2 + r2 < Elo)) {
                if (!amb1) carry1 += r1;
                if (!amb2) carry2 += r2;
                continue;
            }
        }
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const bool in = base + 64 * h + lane < m;
            if (!amb1) {
                const double S = dp_scan_add(d1[h], lane) + carry1;
                const unsigned long long bal = __ballot(in && !(S < Elo));
                if (bal) {  // the reference's loop stops here; it passes the candidate on only if the value is <= eps^2
                    if (kvm_readlane_f64(S, __ffsll((long long)bal) - 1) > Ehi) return DP_REJECT;
                    amb1 = true;
                }
                carry1 = kvm_readlane_f64(S, 63);
            }
            if (!amb2) {
                const double S = dp_scan_add(d2[h], lane) + carry2;
                const unsigned long long bal = __ballot(in && !(S < Elo));
                if (bal) {
                    if (kvm_readlane_f64(S, __ffsll((long long)bal) - 1) > Ehi) return DP_REJECT;
                    amb2 = true;
                }
                carry2 = kvm_readlane_f64(S, 63);
            }
        }
        // A candidate that is still in both loops after the first round (the 256 largest |q|, where the loops of a candidate
        // that is going to be dropped nearly always stop) most likely runs them to the end.  Every term is >= 0, so the
        // partial sums never decrease: if the TOTALS stay below eps^2 (1 - tau), no partial sum reached eps^2 and both loops
        // ran to the end -- and a total needs neither the reference's order nor a scan: all m terms of each bound are summed
        // in NATURAL order, one accumulator per lane and one wave reduction (any order is within tau of the reference's, see
        // kvm_dtw_aux::tau).  Natural order makes every load of the pass coalesced; the ordered rounds gather x[order[i]] and
        // the envelope at order[i], one 64-B sector per 8-B value, and with a few thousand candidates in flight the prologue
        // ran at the L2's transaction rate (0.35 - 0.55 ms per call, round 2).  Only if a total does reach the threshold do
        // the ordered, scanned rounds go on from here to find where.
        DP_STAMP("round");
        if (DP_SUM_MODE && base == 0 && !amb1 && !amb2 && m > 256) {
            double s1 = 0.0, s2 = 0.0;
            // (a window whose data envelope lies wholly inside the fetched chunk -- nearly all of them -- reads the whole-series
            // envelope without a test per point: four terms of each bound per lane and pass, all their loads in flight together)
            const bool interior = c.start - r >= pl.cbeg && c.start + m - 1 + r <= pl.cend;
            if (interior) {
                const double *__restrict__ eu = ax.env_u + (c.start - 1), *__restrict__ el = ax.env_l + (c.start - 1);
                for (int b2 = 0; b2 < m; b2 += 256) {
                    double xv[4], uq[4], lq[4], qv[4], ud[4], ld[4];
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        const int ic = min(b2 + 64 * h + lane, m - 1);
                        xv[h] = x[ic]; uq[h] = qd.qu[ic]; lq[h] = qd.ql[ic]; qv[h] = qd.q[ic]; ud[h] = eu[ic]; ld[h] = el[ic];
                    }
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        const bool in = b2 + 64 * h + lane < m;
                        const double t1 = dp_term_q(xv[h], mean, inv_sd, uq[h], lq[h]);
                        const double t2 = dp_term_d(qv[h], ud[h], ld[h], mean, inv_sd);
                        s1 += in ? t1 : 0.0;
                        s2 += in ? t2 : 0.0;
                    }
                }
            } else {
                for (int b2 = 0; b2 < m; b2 += 256) {
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        const int i = b2 + 64 * h + lane, ic = min(i, m - 1);
                        const double t1 = dp_term_q(x[ic], mean, inv_sd, qd.qu[ic], qd.ql[ic]);
                        double U, L;
                        dp_env_at(ax, patch, r, pl, c.start + ic, U, L);
                        const double t2 = dp_term_d(qd.q[ic], U, L, mean, inv_sd);
                        s1 += i < m ? t1 : 0.0;
                        s2 += i < m ? t2 : 0.0;
                    }
                }
            }
            const double T1 = kvm_wave_sum(s1), T2 = kvm_wave_sum(s2);
            DP_STAMP("totals");
            if (T1 < Elo && T2 < Elo) {
                carry1 = T1;
                carry2 = T2;
                break;
            }
        }
    }
    if (amb1 || amb2) return DP_AMBIG;
    const double S1 = carry1, S2 = carry2;  // (< eps^2 (1 - tau): both loops ran to the end, cb1[] and cb2[] are complete)
    // ---- `if (lbK > lbK2)` (K/NormQueryEngineDtw.java:577): which bound's terms become cb[] ----
    if (S1 == 0.0 && S2 == 0.0) {  // every term of both is zero: cb[] is all zeros either way
        pl.which = 1;
        pl.total = 0.0;
    } else if (fabs(S1 - S2) > ax.tau * fmax(S1, S2)) {
        pl.which = (S1 > S2) ? 0 : 1;
        pl.total = (S1 > S2) ? S1 : S2;
    } else {
        return DP_AMBIG;  // the two sequential sums may compare either way
    }
    return DP_RUN;
}


// The prologue as its own kernel (one wavefront per candidate, grid-stride): it writes one dp_plan per candidate and
// queues the ambiguous ones.  Kept out of the DP kernels so that their register allocation is the sweep's alone (with the
// prologue inlined the register form needed 148 VGPRs -- three waves per SIMD instead of four -- and spilled 100 SGPRs).
__global__ __launch_bounds__(256) void kvm_dtw_prep_kernel(const double *__restrict__ series, kvm_query_dev qd, kvm_dtw_aux ax,
                                                           const kvm_cand *__restrict__ in,
                                                           const unsigned long long *__restrict__ d_n_in,
                                                           dp_plan *__restrict__ plans, unsigned *__restrict__ list,
                                                           unsigned long long *__restrict__ counters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long n_in = *d_n_in;
    const unsigned long long slot = (unsigned long long)blockIdx.x * 4 + wave, slots = (unsigned long long)gridDim.x * 4;
    double *__restrict__ patch = ax.patch + (size_t)slot * 8 * (size_t)qd.rho;
    for (unsigned long long idx = slot; idx < n_in; idx += slots) {
        const kvm_cand c = in[idx];
        dp_plan pl;
        pl.which = DP_PLAN_REJECT; pl.total = 0.0; pl.pad = 0.0; pl.cbeg = pl.cend = pl.sb = 0;
        if (c.start >= 0) {  // (negative: rejected by the exact (alpha, beta) test)
            const int st = dp_prologue(series, qd, ax, c, lane, patch, pl);
            if (st == DP_REJECT) pl.which = DP_PLAN_REJECT;
            if (st == DP_AMBIG) {
                pl.which = DP_PLAN_AMBIG;
                dp_emit_ambig(ax, counters, c, lane);
            }
        }
        if (lane == 0) {
            plans[idx] = pl;
            // the DP kernels take their work from a compact list: every pop is a candidate that runs
            if (pl.which >= 0) list[atomicAdd(&counters[KVM_CNT_DP_LIST], 1ULL)] = (unsigned)idx;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// DP, generic form (any rho): banded DP swept by anti-diagonals, one wavefront per candidate, band rows in LDS.
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
// Row i is complete after diagonal i + min(m-1, i+r); the rows the reference tests (i + r < m - 1) complete on diagonal
// 2i + r, in increasing order of i, so the first row whose bound reaches eps^2 is found in the reference's order.
extern __shared__ double kvm_dp_lds[];

// doubles per candidate: band rows E, O (r+1 each), window A, cb[] and row minima (m each), envelope patch (8r)
__host__ __device__ static inline size_t dp_lds_doubles(int m, int r) { return (size_t)2 * (r + 1) + (size_t)3 * m + (size_t)8 * r; }

// `scratch` != nullptr: the arrays do not fit the LDS (m + 2 rho > 6 000 or so) and live in a per-workgroup slice of
// global memory instead -- same code, one wavefront per workgroup, so the workgroup barrier (a workgroup-scope fence)
// orders its stores and loads through the CU's own L1.
template <bool ABANDON>
__global__ __launch_bounds__(64) void kvm_dtw_dp_lds_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                            kvm_dtw_aux ax, const kvm_cand *__restrict__ in,
                                                            const dp_plan *__restrict__ plans,
                                                            const unsigned long long *__restrict__ d_n_in,
                                                            int32_t *__restrict__ out_off,
                                                            double *__restrict__ out_dist, int64_t cap,
                                                            unsigned long long *__restrict__ counters,
                                                            double *scratch) {
    const int lane = threadIdx.x;
    const int m = qd.m, r = qd.rho;
    const double eps2 = qd.eps2;
    const double *__restrict__ B = qd.q;
    double *E = scratch ? scratch + (size_t)blockIdx.x * dp_lds_doubles(m, r) : kvm_dp_lds;  // r+1 entries
    double *O = E + (r + 1);     // r+1 entries (only r used)
    double *A = O + (r + 1);     // m entries: the (z-normalised) candidate window
    double *CB = A + m;          // m entries: cb[] (K/NormQueryEngineDtw.java:576-587)
    double *RM = CB + m;         // m entries: row minima (min_cost of K/utils/DtwUtils.java:286,319-321)
    double *patch = RM + m;      // 8r entries: data envelope at the chunk ends
    const unsigned long long n_in = *d_n_in;
    const bool exact = ax.cb_exact != nullptr;
    const double tol = 4.0 * ax.tau * eps2;

    // (one pop per iteration, at most n_in successful pops in the whole grid: the bound is a guard, every wave leaves)
    for (unsigned long long pops = 0; pops <= n_in; pops++) {
        unsigned long long popped = 0;
        if (lane == 0) popped = atomicAdd(&counters[KVM_CNT_QUEUE], 1ULL);
        const unsigned long long idx = kvm_uniform_u64(popped);  // (a scalar: see kvm_uniform_u64)
        if (idx >= n_in) break;
        const kvm_cand c = in[idx];
        if (c.start < 0) continue;  // rejected by the exact (alpha, beta) test / by the sequential path
        const double *__restrict__ x = series + (c.start - 1);
        dp_plan pl;
        pl.which = 0; pl.total = 0.0; pl.pad = 0.0; pl.cbeg = pl.cend = pl.sb = 0;
        if (!exact) {
            pl = plans[idx];
            if (pl.which < 0) continue;  // dropped by the prologue, or queued for the sequential path
            if (pl.which == 1) dp_build_patch(series, r, pl, c.start, m, lane, patch);
        }
        if (lane == 0) atomicAdd(&counters[KVM_CNT_PASS_KEOGH2], 1ULL);
        // cb[]: exact arrays of the sequential path, or total - prefix of the chosen bound's terms
        if (exact) {
            const double *__restrict__ src = ax.cb_exact + (size_t)idx * 2 * m + (c.lim != 0.0 ? m : 0);
            for (int k = lane; k < m; k += 64) CB[k] = src[k];
        } else {
            double pref = 0.0;
            for (int base = 0; base < m; base += 64) {
                const int k = base + lane;
                const double t = k < m ? dp_cb_term(x, qd, ax, patch, pl, c.start, c.mean, c.std, k) : 0.0;
                const double inc = dp_scan_add(t, lane);
                if (k < m) CB[k] = pl.total - (pref + (inc - t));
                pref += kvm_readlane_f64(inc, 63);
            }
        }
        // zT[k] = (T[k+j] - mean) / std   (K/NormQueryEngineDtw.java:564-567); RSM: mean 0, std 1 -> exact copy
        for (int i = lane; i < m; i += 64) {
            A[i] = (x[i] - c.mean) / c.std;
            RM[i] = DTW_INF;
        }
        for (int e = lane; e <= r; e += 64) {
            E[e] = DTW_INF;
            O[e] = DTW_INF;
        }
        __syncthreads();
        int decision = 0;  // 0 none yet, 1 reject, 2 ambiguous, 3 the reference returns `value` at a row
        double value = 0.0;
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
                        if (v < RM[i]) RM[i] = v;  // (one cell per row on a diagonal: no two lanes share a row)
                    }
                }
            } else {
                for (int o = lane; o <= r - 1; o += 64) {
                    const int i = (d + r - 2 * o - 1) >> 1, j = d - i;
                    if (d + r - 2 * o - 1 >= 0 && i < m && j >= 0 && j < m) {
                        const double dd = dsq(A[i], B[j]);
                        double v;
                        if (i == 0 && j == 0) {  // r odd: the origin is an odd-k cell
                            v = dd;
                        } else {
                            const double xx = E[o + 1];
                            const double yy = E[o];
                            v = fmin(fmin(xx, yy), O[o]) + dd;
                        }
                        O[o] = v;
                        if (v < RM[i]) RM[i] = v;
                    }
                }
            }
            __syncthreads();
            // row i = (d - r) / 2 has just received its last cell (i, i + r): `if (i + r < m - 1 && min_cost + cb[i+r+1] >= bsf)`
            if (decision == 0 && d >= r && ((d - r) & 1) == 0) {
                const int i = (d - r) >> 1;
                if (i + r < m - 1) {
                    const int t = dp_row_test(lane == 0, RM[i], CB[i + r + 1], eps2, tol, exact, value);
                    if (t) {
                        decision = t;
                        if (ABANDON) break;
                    }
                }
            }
        }
        __syncthreads();
        if (decision == 2) {
            dp_emit_ambig(ax, counters, c, lane);
            if (lane == 0) atomicAdd(&counters[KVM_CNT_PASS_KEOGH2], ~0ULL);  // (counted again when the exact run starts it)
        } else if (decision != 1 && lane == 0) {
            // no row reached eps^2: cell (m-1, m-1), k = r (:333-336); else the row bound the reference returned
            const double v = decision == 3 ? value : ((r & 1) ? O[(r - 1) >> 1] : E[r >> 1]);
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
// DP, register form: the same red/black anti-diagonal sweep with the whole band in VGPRs.
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
//
// Row minima (min_cost, :286,319-321).  The E and O cells a slot (lane p, index c) updates in macro-step I both belong
// to row I - pV - c, and that row moves on to the next slot every macro-step -- exactly like the data value a[].  So the
// minimum of a row travels with it: a register per slot under the same static renaming as a[], handed to the next lane
// by one DPP move when the row leaves a lane's last slot (lane 0 starts every new row at INF), two v_min_f64 per slot
// and macro-step.  Row i receives its last cell, E[r] (k = 2r), in macro-step i + r, in the lane / slot that also
// holds the band-edge cell: that lane stores the finished minimum into a 64-entry LDS strip, and after every 64
// macro-steps the wavefront tests the 64 finished rows in parallel (dp_row_test) -- in row order, so the first row
// whose bound reaches eps^2 is the reference's.  The cb[] values the test needs are formed right there: total - prefix
// of the chosen bound's terms, one term per lane and chunk.

// LDS window buffers of the register-form DP.  The data/query values a lane needs advance by one index per macro-step:
// a block of V macro-steps reads the V values [j V, (j + 1) V) of each window, j = j0 -+ lane, and j0 advances by one per
// block.  A buffer holds the 64*V values behind the front plus DP_EPOCH chunks of 64 new values; after DP_EPOCH chunks
// the live tail is slid back to the front.  Layout (round 3): a lane's V values are fetched with V/2 ds_read_b128, and for
// those to be free of bank conflicts consecutive lanes must hit consecutive 16-byte slots -- so the PAIRS (2h, 2h+1) of
// every group of V go to sub-buffer h, group j at doubles [2j, 2j+2) of it: lanes j0 -+ lane then read one contiguous
// kilobyte per instruction.  (The linear layout of rounds 1-2 read with a lane stride of 8V bytes: 57 % of the LDS cycles
// of the kernel were bank conflicts.)  Values are fetched one block AHEAD of their use into a second register set.
#ifndef DP_EPOCH
#define DP_EPOCH 4
#endif
#ifndef DP_BLOCKS_PER_CU
#define DP_BLOCKS_PER_CU 4
#endif
template <int V>
struct dp_buf {
    static constexpr int LIVE = 64 * V;                  // values behind the front that lanes still read
    static constexpr int LIN = LIVE + 64 * DP_EPOCH;     // doubles per buffer (V=4: 512 = 4 KiB)
    static constexpr int NSUB = V >= 2 ? V / 2 : 1;
    static constexpr int SUB = LIN / NSUB;               // doubles per sub-buffer
    __device__ static __forceinline__ int phys(int p) {  // where linear position p >= 0 lives
        if constexpr (V <= 2) return p;
        else return ((p % V) >> 1) * SUB + (p / V) * 2 + (p & 1);
    }
    // the V values of a group: `at` = address of the group's first pair in sub-buffer 0 (group j of a buffer: buf + GROUP * j
    // doubles), `ahead` groups further on (an immediate offset)
    static constexpr int GROUP = V == 1 ? 1 : 2;
    __device__ static __forceinline__ void load(const double *at, int ahead, double (&out)[V]) {
        if constexpr (V == 1) {
            out[0] = at[ahead];
        } else {
#pragma unroll
            for (int h = 0; h < V / 2; h++) {
                const double2 t = *reinterpret_cast<const double2 *>(at + h * SUB + 2 * ahead);
                out[2 * h] = t.x;
                out[2 * h + 1] = t.y;
            }
        }
    }
};


// A candidate is swept with its row minima on high words first; if the first row that may reach eps^2 cannot be told from
// a high word (about one candidate in a thousand: the bound of successive rows creeps up on eps^2, so the first row past it
// is often within 2^-20 of it), the same wave sweeps it again at once with the minima in full -- in the kernel, so that the
// second sweep hides among the other waves' work instead of adding a launch of lone DPs behind the first.  What is then
// still within the tolerance of the approximate cb[] goes to the sequential path (ax.ambig).  EXACT: the run on the
// sequential path's candidates -- exact cb[] (ax.cb_exact), minima in full, the reference's own comparison.
template <int V, int CSTAR, bool ABANDON, bool EXACT>
__device__ __forceinline__ void dp_reg_sweep(const double *__restrict__ series, const kvm_query_dev &qd, const kvm_dtw_aux &ax,
                                             const kvm_cand *__restrict__ in, const dp_plan *__restrict__ plans,
                                             const unsigned long long *__restrict__ d_n_in, int32_t *__restrict__ out_off,
                                             double *__restrict__ out_dist, int64_t cap,
                                             unsigned long long *__restrict__ counters, const unsigned *__restrict__ list,
                                             int share_mode, int num_cu) {
    using BUF = dp_buf<V>;
    constexpr int LIN = BUF::LIN, LIVE = BUF::LIVE, NBLK = 64 / V;
    constexpr int WAVE_LDS = 2 * LIN + 64 * V + 64;  // buffer A, buffer B, result scratch, finished row minima
    __shared__ __attribute__((aligned(16))) double lds_all[4 * WAVE_LDS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *bufA = lds_all + wave * WAVE_LDS, *bufB = bufA + LIN, *scratch = bufB + LIN, *bufR = scratch + 64 * V;
    unsigned *bufR32 = reinterpret_cast<unsigned *>(bufR);
    char *lds_raw = reinterpret_cast<char *>(lds_all);
    const int m = qd.m, r = qd.rho;
    const double eps2 = qd.eps2;
    const double *__restrict__ Bq = qd.q;
    // first sweep: the entries of the prologue's compact list behind the packed kernel's share; EXACT: the whole of `in`
    const unsigned long long n_list = EXACT ? 0ULL : counters[KVM_CNT_DP_LIST];
    const unsigned long long first = EXACT ? 0ULL : dp_pack_share(n_list, num_cu, share_mode, r);
    const unsigned long long n_in = EXACT ? *d_n_in : n_list - first;
    const int I0 = r >> 1, Ilast = m - 1 + I0;
    const int pV = lane * V;
    const bool edge_lane = lane == r / V;  // the band-edge cell O[r] is cell CSTAR (= r % V) of this lane
    const int kfin = (r & 1) ? ((r - 1) >> 1) : (r >> 1);  // index of cell (m-1, m-1) (k = r) in O (r odd) / E (r even)
    const double tol = 4.0 * ax.tau * eps2;
    const unsigned INF_HI = (unsigned)__double2hiint(DTW_INF);
    double *__restrict__ patch = ax.patch + (size_t)(blockIdx.x * 4 + wave) * 8 * (size_t)r;

    // (one pop per iteration, at most n_in successful pops in the whole grid: the bound is a guard, every wave leaves)
    for (unsigned long long pops = 0; pops <= n_in; pops++) {
        // (a look before the pop: when this kernel has nothing to do -- the packed kernel took the whole list -- or the queue is
        // drained, its thousands of waves would otherwise each queue one atomic on the same word: 30 - 40 us of nothing)
        if (__hip_atomic_load(&counters[EXACT ? KVM_CNT_QUEUE : KVM_CNT_QUEUE2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_in) break;
        unsigned long long popped = 0;
        if (lane == 0) popped = atomicAdd(&counters[EXACT ? KVM_CNT_QUEUE : KVM_CNT_QUEUE2], 1ULL);
        const unsigned long long pos = kvm_uniform_u64(popped);  // (a scalar: see kvm_uniform_u64)
        if (pos >= n_in) break;
        const unsigned long long idx = EXACT ? pos : (unsigned long long)list[first + pos];
        const kvm_cand cand = in[idx];
        if (cand.start < 0) continue;  // rejected by the exact (alpha, beta) test / by the sequential path
        const double *__restrict__ x = series + (cand.start - 1);
        const double mean = cand.mean, sd = cand.std;
        dp_plan pl;
        pl.which = 0; pl.total = 0.0; pl.pad = 0.0; pl.cbeg = pl.cend = pl.sb = 0;
        if (!EXACT) {
            pl = plans[idx];
            if (pl.which < 0) continue;  // dropped by the prologue, or queued for the sequential path
            if (pl.which == 1) dp_build_patch(series, r, pl, cand.start, m, lane, patch);
        }
        if (lane == 0) atomicAdd(&counters[KVM_CNT_PASS_KEOGH2], 1ULL);
        const double *__restrict__ cb_src = EXACT ? ax.cb_exact + (size_t)idx * 2 * m + (cand.lim != 0.0 ? m : 0) : nullptr;
        // prefix of the cb terms below the first index the row test needs (cb[I0 + 1]: chunk Ic covers cb[Ic+1 .. Ic+64])
        double pref0 = 0.0;
        if (!EXACT) {
            for (int base = 0; base <= I0; base += 64) {
                const int k = base + lane;
                pref0 += kvm_wave_sum((k <= I0 && k < m) ? dp_cb_term(x, qd, ax, patch, pl, cand.start, mean, sd, k) : 0.0);
            }
        }
        double result = DTW_INF, value = 0.0;
        // one sweep of the candidate; full_tag: row minima as doubles (true) or as high words (false).  Returns the decision:
        // 0 none (`result` = cell (m-1, m-1)), 1 reject, 2 cannot tell, 3 the reference returns `value` at a row (EXACT)
        auto one = [&](auto full_tag) -> int {
        constexpr bool FULLMIN = decltype(full_tag)::value;
        double pref = pref0;
        // linear position p of a buffer holds A[lowA + p] / B[lowB + p] (stored at BUF::phys(p))
        int lowA = I0 - LIVE, lowB = I0 - r;
        // zT[k] = (T[k+j] - mean) / std (K/NormQueryEngineDtw.java:564-567); RSM: (x - 0) / 1 == x exactly
        auto fillA = [&](int pos, int cnt) {
            for (int t = lane; t < cnt; t += 64) {
                const int i = lowA + pos + t;
                bufA[BUF::phys(pos + t)] = (i >= 0 && i < m) ? (x[i] - mean) / sd : DTW_POS_BIG;
            }
        };
        auto fillB = [&](int pos, int cnt) {
            for (int t = lane; t < cnt; t += 64) {
                const int j = lowB + pos + t;
                bufB[BUF::phys(pos + t)] = (j >= 0 && j < m) ? Bq[j] : DTW_NEG_BIG;
            }
        };
        fillA(0, LIVE + 64);
        fillB(0, LIVE + 64);

        double E[V], O[V], a[V], b[V], R[FULLMIN ? V : 1];
        unsigned Rh[FULLMIN ? 1 : V];
#pragma unroll
        for (int c = 0; c < V; c++) {
            E[c] = DTW_INF;
            O[c] = DTW_INF;
            if constexpr (FULLMIN) R[c] = DTW_INF;
            else Rh[c] = INF_HI;
            a[c] = c >= 1 ? bufA[BUF::phys(I0 - pV - c - lowA)] : 0.0;
            b[c] = bufB[BUF::phys(I0 - r + pV + c - lowB)];
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
        // read groups: the block of macro-steps I .. I+V-1 reads A[I - pV ..] and B[I - r + pV + V ..], i.e. group
        // ja = (I - pV - lowA) / V of buffer A and jb = (I - r + pV + V - lowB) / V of buffer B (both exact: multiples of V)
        const double *ja = bufA + BUF::GROUP * ((LIVE - pV) / V), *jb = bufB + BUF::GROUP * (lane + 1);
        // halos: lane 0 (left) / lane 63 (right) never receive from a neighbour, so they keep this initial INF for ever
        double halo_l = DTW_INF, halo_r = DTW_INF, r_in = DTW_INF;
        unsigned rh_in = INF_HI;
        const double edge_pen = edge_lane ? DTW_INF : 0.0;
        bool done = false;
        int decision = 0;  // 0 none yet, 1 reject, 2 ambiguous, 3 the reference returns `value` at a row
        int chunk = 0;

        // one block of V macro-steps, from macro-step Ib, with the window values (an, bn) fetched for it; CHECKED: the block
        // that holds macro-step Ilast (cell (m-1, m-1) is read out of it)
        auto block = [&](auto checked, const double(&an)[V], const double(&bn)[V], const int Ib, const unsigned pr) {
            constexpr bool CHECKED = decltype(checked)::value;
#pragma unroll
            for (int s = 0; s < V; s++) {
                const int I = Ib + s;
                // the row that left this lane's last slot moves on to the next lane; lane 0 starts a new row: it
                // never receives, so its r_in keeps the INF it was initialised with (no constant to re-materialise)
                if constexpr (FULLMIN) r_in = dp_from_left(R[(V - s) % V], r_in);
                else rh_in = dp_u32_from_left(Rh[(V - s) % V], rh_in);
                // ---- E-update (diagonal 2I - r) ----
                halo_l = dp_from_left(O[V - 1], halo_l);
#pragma unroll
                for (int c = V - 1; c >= 0; c--) {
                    if (c == 0) a[(V - s) % V] = an[s];  // slot of cell 0 at step s
                    const double av = a[((c - s) % V + V) % V], bv = b[(c + s) % V];
                    const double dd = (av - bv) * (av - bv);
                    const double yy = c > 0 ? O[c - 1] : halo_l;
                    E[c] = dp_min(dp_min(O[c], yy), E[c]) + dd;
                }
                if (CHECKED && (r & 1) == 0 && I == Ilast) {  // r even: cell (m-1, m-1) is E[r/2] of this update
#pragma unroll
                    for (int c = 0; c < V; c++) scratch[pV + c] = E[c];
                    result = scratch[kfin];
                    done = true;
                }
                // ---- O-update (diagonal 2I - r + 1) ----
                halo_r = dp_from_right(E[0], halo_r);
                b[s % V] = bn[s];  // slot (V + s) % V: query value of O cell V-1, E cell V-1 of the next step
#pragma unroll
                for (int c = 0; c < V; c++) {
                    const double av = a[((c - s) % V + V) % V], bv = b[(c + 1 + s) % V];
                    const double dd = (av - bv) * (av - bv);
                    const double xx = c < V - 1 ? E[c + 1] : halo_r;
                    O[c] = dp_min(dp_min(xx, E[c]), O[c]) + dd;
                }
                if (CHECKED && (r & 1) == 1 && I == Ilast) {
#pragma unroll
                    for (int c = 0; c < V; c++) scratch[pV + c] = O[c];
                    result = scratch[kfin];
                    done = true;
                }
                // band edge k = 2r+1: O[r] must read as INF (>= 1e20: `edge_pen` is 1e20 in that lane, 0 elsewhere)
                O[CSTAR] += edge_pen;
                // ---- row minima (slot 0 takes over the row handed in from the left) ----
                if constexpr (FULLMIN) {
#pragma unroll
                    for (int c = 0; c < V; c++)
                        R[((c - s) % V + V) % V] = dp_min(dp_min(c == 0 ? r_in : R[((c - s) % V + V) % V], E[c]), O[c]);
                    if (edge_lane) *reinterpret_cast<double *>(lds_raw + pr + 8 * s) = R[((CSTAR - s) % V + V) % V];  // row I - r is complete
                } else {
#pragma unroll
                    for (int c = 0; c < V; c++)
                        Rh[((c - s) % V + V) % V] = dp_umin3(c == 0 ? rh_in : Rh[((c - s) % V + V) % V],
                                                             (unsigned)__double2hiint(E[c]), (unsigned)__double2hiint(O[c]));
                    if (edge_lane) *reinterpret_cast<unsigned *>(lds_raw + pr + 4 * s) = Rh[((CSTAR - s) % V + V) % V];
                }
            }
        };

        for (int Ic = I0; Ic <= Ilast && !done; Ic += 64) {
            // blocks of this chunk that lie wholly before macro-step Ilast run unchecked; the one holding Ilast, if it is in
            // this chunk, is the last block of the candidate
            const int nb = min(NBLK, (Ilast - Ic) / V);
            // Register sets of window values: the block in flight computes from set u while the next block's values land in
            // set u+1 -- and the values of the block before it, which the slots a[] / b[] still hold, stay where they were
            // fetched (set u-1): three sets in rotation, no copies.  (V = 8 has no registers to spare: one set, fetched
            // at the top of its block.)
            constexpr int NSETS = V <= 4 ? 3 : 1;
            double as[NSETS][V], bs[NSETS][V];
            BUF::load(ja, 0, as[0]);   // (the first block of a chunk waits for its values: they were filled a moment ago)
            BUF::load(jb, 0, bs[0]);
            int blk = 0;
            // byte offset of the strip of finished row minima, in a VGPR of its own (a uniform address would be re-materialised
            // from an SGPR in front of every masked store)
            unsigned pr = (unsigned)((char *)bufR - lds_raw);
            asm volatile("" : "+v"(pr));
            constexpr unsigned PR_STEP = V * (FULLMIN ? 8 : 4);
            bool more = blk < nb;
#pragma unroll 1
            while (more) {
#pragma unroll
                for (int u = 0; u < NSETS; u++) {
                    if (more) {
                        if constexpr (NSETS > 1) {
                            BUF::load(ja, 1, as[(u + 1) % NSETS]);   // (the group after the chunk's last is fetched and
                            BUF::load(jb, 1, bs[(u + 1) % NSETS]);   //  dropped: it lies inside this wave's LDS region)
                        } else if (blk > 0) {
                            BUF::load(ja, 0, as[0]);
                            BUF::load(jb, 0, bs[0]);
                        }
                        block(std::false_type{}, as[u], bs[u], Ic + blk * V, pr);
                        blk++; ja += BUF::GROUP; jb += BUF::GROUP; pr += PR_STEP;
                        more = blk < nb;
                    }
                }
            }
            if (nb < NBLK) {
                BUF::load(ja, 0, as[0]);
                BUF::load(jb, 0, bs[0]);
                block(std::true_type{}, as[0], bs[0], Ic + nb * V, pr);
                ja += BUF::GROUP; jb += BUF::GROUP;
            }
            // ---- row test of the rows finished in macro-steps Ic .. Ic+63 (lane t: row Ic + t - r) ----
            if (decision == 0) {
                const int I = Ic + lane, k = I + 1;  // needs cb[I + 1]
                double cbv, term = 0.0;
                if (EXACT) {
                    cbv = (k < m) ? cb_src[k] : 0.0;
                } else {
                    term = (k < m) ? dp_cb_term(x, qd, ax, patch, pl, cand.start, mean, sd, k) : 0.0;
                    const double inc = dp_scan_add(term, lane);
                    cbv = pl.total - (pref + (inc - term));
                    pref += kvm_readlane_f64(inc, 63);
                }
                const bool valid = I >= r && I <= m - 2;
                const int t = FULLMIN ? dp_row_test(valid, bufR[lane], cbv, eps2, tol, EXACT, value)
                                      : dp_row_test_hi(valid, bufR32[lane], cbv, eps2, tol);
                if (t) {
                    decision = t;
                    if (ABANDON) break;
                }
            }
            if (done) break;
            chunk++;
            if (chunk == DP_EPOCH) {
                // slide: the LIVE values behind the front move to the start of the buffers
                double ta[V], tb[V];
#pragma unroll
                for (int c = 0; c < V; c++) {
                    ta[c] = bufA[BUF::phys(64 * DP_EPOCH + lane + 64 * c)];
                    tb[c] = bufB[BUF::phys(64 * DP_EPOCH + lane + 64 * c)];
                }
#pragma unroll
                for (int c = 0; c < V; c++) {
                    bufA[BUF::phys(lane + 64 * c)] = ta[c];
                    bufB[BUF::phys(lane + 64 * c)] = tb[c];
                }
                lowA += 64 * DP_EPOCH;
                lowB += 64 * DP_EPOCH;
                ja -= BUF::GROUP * (64 * DP_EPOCH / V);
                jb -= BUF::GROUP * (64 * DP_EPOCH / V);
                chunk = 0;
            }
            // the 64 values the next chunk adds at the front of each window
            fillA(LIVE + 64 * chunk, 64);
            fillB(LIVE + 64 * chunk, 64);
        }
        return decision;
        };
        int decision;
        if constexpr (EXACT) {
            decision = one(std::true_type{});
        } else {
            decision = one(std::false_type{});
            if (decision == 2) {
                if (lane == 0) atomicAdd(&counters[KVM_CNT_DP_RERUN], 1ULL);
                decision = one(std::true_type{});
            }
        }
        if (decision == 2) {
            dp_emit_ambig(ax, counters, cand, lane);
            if (lane == 0) atomicAdd(&counters[KVM_CNT_PASS_KEOGH2], ~0ULL);  // (counted again when the exact run starts it)
        } else if (decision != 1 && lane == 0) {
            const double v = decision == 3 ? value : result;  // (:325 the row bound / :333-336 the last cell)
            if (v <= eps2) {
                unsigned long long o = atomicAdd(&counters[KVM_CNT_ANSWERS], 1ULL);
                if ((int64_t)o < cap) {
                    out_off[o] = cand.start;
                    out_dist[o] = sqrt(v);
                }
            }
        }
    }
}

// The kernels.  V <= 4 (rho <= 255): held to 168 VGPRs, three waves per SIMD -- the workgroup's LDS (42 KiB at V = 4)
// allows three workgroups per CU, and the third wave fills the issue slots the other two leave at s_nop / s_waitcnt
// (3.87 -> 4.16e12 cells/s on 16 384 full DPs, profiles/r03_ab_dtw.log).  V = 8 needs 224 VGPRs: two waves.
template <int V, int CSTAR, bool ABANDON, bool EXACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void kvm_dtw_dp_reg_kernel(
    const double *__restrict__ series, kvm_query_dev qd, kvm_dtw_aux ax, const kvm_cand *__restrict__ in,
    const dp_plan *__restrict__ plans, const unsigned long long *__restrict__ d_n_in, int32_t *__restrict__ out_off,
    double *__restrict__ out_dist, int64_t cap, unsigned long long *__restrict__ counters, const unsigned *__restrict__ list,
    int share_mode, int num_cu) {
    dp_reg_sweep<V, CSTAR, ABANDON, EXACT>(series, qd, ax, in, plans, d_n_in, out_off, out_dist, cap, counters, list, share_mode, num_cu);
}
template <int V, int CSTAR, bool ABANDON, bool EXACT>
__global__ __launch_bounds__(256) void kvm_dtw_dp_reg_wide_kernel(
    const double *__restrict__ series, kvm_query_dev qd, kvm_dtw_aux ax, const kvm_cand *__restrict__ in,
    const dp_plan *__restrict__ plans, const unsigned long long *__restrict__ d_n_in, int32_t *__restrict__ out_off,
    double *__restrict__ out_dist, int64_t cap, unsigned long long *__restrict__ counters, const unsigned *__restrict__ list,
    int share_mode, int num_cu) {
    dp_reg_sweep<V, CSTAR, ABANDON, EXACT>(series, qd, ax, in, plans, d_n_in, out_off, out_dist, cap, counters, list, share_mode, num_cu);
}

template <int V, int CSTAR>
static hipError_t launch_dp_reg_c(const double *series, const kvm_query_dev &qd, const kvm_dtw_aux &ax, const kvm_cand *in,
                                  const dp_plan *plans, const unsigned long long *d_n_in, int64_t max_in, int32_t *out_off, double *out_dist,
                                  int64_t cap, unsigned long long *counters, int num_cu, const unsigned *list, int share_mode,
                                  hipStream_t s) {
    // resident workgroups per CU = waves per SIMD: bounded by the LDS (4 waves x (2*LIN + 64V + 64) doubles per workgroup)
    // and by the registers (three waves for V <= 4, two for V = 8, see the kernels)
    const int lds_limit = (int)((160 * 1024) / (4 * (2 * dp_buf<V>::LIN + 64 * V + 64) * sizeof(double)));
    const int reg_limit = V <= 4 ? 3 : 2;
    int64_t blocks = (int64_t)num_cu * (reg_limit < lds_limit ? reg_limit : lds_limit);
    if (blocks > (max_in + 3) / 4) blocks = (max_in + 3) / 4;
#define KVM_DP_LAUNCH(AB, EX)                                                                                               \
    do {                                                                                                                     \
        if constexpr (V <= 4)                                                                                                \
            hipLaunchKernelGGL((kvm_dtw_dp_reg_kernel<V, CSTAR, AB, EX>), dim3((unsigned)blocks), dim3(256), 0, s, series,   \
                               qd, ax, in, plans, d_n_in, out_off, out_dist, cap, counters, list, share_mode, num_cu);        \
        else                                                                                                                 \
            hipLaunchKernelGGL((kvm_dtw_dp_reg_wide_kernel<V, CSTAR, AB, EX>), dim3((unsigned)blocks), dim3(256), 0, s,      \
                               series, qd, ax, in, plans, d_n_in, out_off, out_dist, cap, counters, list, share_mode, num_cu); \
    } while (0)
    const bool exact = ax.cb_exact != nullptr;  // the run on the sequential path's candidates: exact cb[], the reference's comparison
    if (qd.early_abandon) {
        if (exact) KVM_DP_LAUNCH(true, true);
        else KVM_DP_LAUNCH(true, false);
    } else {
        if (exact) KVM_DP_LAUNCH(false, true);
        else KVM_DP_LAUNCH(false, false);
    }
#undef KVM_DP_LAUNCH
    return hipGetLastError();
}

// dispatch on the (launch-constant) position CSTAR = rho % V of the band-edge cell
template <int V>
static hipError_t launch_dp_reg(const double *series, const kvm_query_dev &qd, const kvm_dtw_aux &ax, const kvm_cand *in,
                                const dp_plan *plans, const unsigned long long *d_n_in, int64_t max_in, int32_t *out_off, double *out_dist,
                                int64_t cap, unsigned long long *counters, int num_cu, const unsigned *list, int share_mode,
                                hipStream_t s) {
#define KVM_DP_CASE(C)                                                                                          \
    case C:                                                                                                     \
        if constexpr (C < V)                                                                                    \
            return launch_dp_reg_c<V, C>(series, qd, ax, in, plans, d_n_in, max_in, out_off, out_dist, cap, counters, num_cu, list, share_mode, s); \
        break;
    switch (qd.rho % V) {
        KVM_DP_CASE(0) KVM_DP_CASE(1) KVM_DP_CASE(2) KVM_DP_CASE(3)
        KVM_DP_CASE(4) KVM_DP_CASE(5) KVM_DP_CASE(6) KVM_DP_CASE(7)
    }
#undef KVM_DP_CASE
    return hipErrorInvalidValue;
}

int64_t kvm_dtw_dp_max_waves(int num_cu) { return (int64_t)num_cu * 4 * 2 * DP_BLOCKS_PER_CU; }  // (the prep kernel's grid)

size_t kvm_dtw_plan_bytes(int64_t n) { return (size_t)n * (sizeof(dp_plan) + sizeof(unsigned)); }

// `plan_buf`: kvm_dtw_plan_bytes(max_in) bytes of scratch for the per-candidate plan records (unused when ax.cb_exact is
// set: the second run on the sequential path's candidates has nothing left to decide in front of the DP)
hipError_t kvm_launch_dtw_dp(const double *series, int64_t n, const kvm_query_dev &qd, const kvm_dtw_aux &ax, const kvm_cand *in,
                             const unsigned long long *d_n_in, int64_t max_in, void *plan_buf, int32_t *out_off, double *out_dist,
                             int64_t cap, unsigned long long *counters, int num_cu, int dp_form, hipStream_t s,
                             hipEvent_t ev_dp_start) {
    (void)n;
    if (max_in <= 0) return hipSuccess;
    dp_plan *plans = reinterpret_cast<dp_plan *>(plan_buf);
    unsigned *list = ax.cb_exact ? nullptr : reinterpret_cast<unsigned *>(plans + max_in);
    if (!ax.cb_exact) {
        int64_t pblocks = (int64_t)num_cu * 2 * DP_BLOCKS_PER_CU;  // (as many waves as the patch scratch has slices)
        if (pblocks > (max_in + 3) / 4) pblocks = (max_in + 3) / 4;
        hipLaunchKernelGGL(kvm_dtw_prep_kernel, dim3((unsigned)pblocks), dim3(256), 0, s, series, qd, ax, in, d_n_in, plans,
                           list, counters);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (ev_dp_start) {  // (kernel timing: the DP alone, as rocprofv3 lists it)
        const hipError_t e = hipEventRecord(ev_dp_start, s);
        if (e != hipSuccess) return e;
    }
    const bool force_lds = dp_form == 1;
    // packed register form (kvm_dtw_pair.hip): any V, two candidates per wave for rho <= 255.  dp_pack_share gives it either the
    // whole compact list or nothing (a function of rho and the option alone): exactly one of the two register kernels is launched.
    if (dp_form != 1 && dp_form != 2 && !ax.cb_exact && qd.m >= 2 && dp_pack_share(1ULL, num_cu, dp_form, qd.rho) > 0)
        return kvm_launch_dtw_dp_pack(series, qd, ax, in, plans, list, dp_form, max_in, out_off, out_dist, cap, counters, num_cu, s);
    // register form: band of r+1 even-k cells over 64 lanes, V = pow2 cells per lane, V <= 8 (r <= 511)
    const int need = (qd.rho + 1 + 63) / 64;
    if (!force_lds && need <= 8 && qd.m >= 2) {
        if (need <= 1) return launch_dp_reg<1>(series, qd, ax, in, plans, d_n_in, max_in, out_off, out_dist, cap, counters, num_cu, list, dp_form, s);
        if (need <= 2) return launch_dp_reg<2>(series, qd, ax, in, plans, d_n_in, max_in, out_off, out_dist, cap, counters, num_cu, list, dp_form, s);
        if (need <= 4) return launch_dp_reg<4>(series, qd, ax, in, plans, d_n_in, max_in, out_off, out_dist, cap, counters, num_cu, list, dp_form, s);
        return launch_dp_reg<8>(series, qd, ax, in, plans, d_n_in, max_in, out_off, out_dist, cap, counters, num_cu, list, dp_form, s);
    }
    // generic form (any r): arrays in LDS when they fit, else in a stream-ordered global scratch
    const size_t need_bytes = dp_lds_doubles(qd.m, qd.rho) * sizeof(double);
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
        hipLaunchKernelGGL((kvm_dtw_dp_lds_kernel<true>), dim3((unsigned)blocks), dim3(64), lds, s, series, qd, ax, in,
                           plans, d_n_in, out_off, out_dist, cap, counters, scratch);
    else
        hipLaunchKernelGGL((kvm_dtw_dp_lds_kernel<false>), dim3((unsigned)blocks), dim3(64), lds, s, series, qd, ax, in,
                           plans, d_n_in, out_off, out_dist, cap, counters, scratch);
    if (scratch) {
        const hipError_t e1 = hipGetLastError(), e2 = hipFreeAsync(scratch, s);
        return e1 != hipSuccess ? e1 : e2;
    }
    return hipGetLastError();
}
