// kvm_ed_dot.hip -- correlation-form ED pruner (RSM-ED, cNSM-ED) for gfx950: ONE v_fma_f64 per (candidate, point).
//
// The exact pass (kvm_exact.hip) decides membership and computes the reference's distance for every survivor, so the
// pruning kernel may evaluate the distance in any algebraically equivalent form as long as it never drops a
// candidate whose true value is within the rounding uncertainty of the threshold.  This kernel uses
//     RSM : sum (x-q)^2            = S2' - 2 (acc + d S1') + T2 + 2 d T1 + m d^2
//     cNSM: sum ((x-mu)/sg - zq)^2 = sum (x-mu)^2 / sg^2 - (2/sg) sum (x-mu) zq + sum zq^2
// with  x' = x - c (c = first point of the tile: removes the level of the data), qc = q - mean(q),
//       acc = sum x'_j qc_j  (the only O(m) work: one FMA per term),
//       S1' = sum x', S2' = sum x'^2 (O(1) from the shifted block prefix sums), T1 = sum qc, T2 = sum qc^2, d = mean(q) - c.
// Everything is formed around local levels, so the terms are O(m var) and the cancellation the naive dot-product
// identity suffers from is bounded by an explicit guard band that is a FUNCTION OF THE NUMBER OF TERMS
// (edd_dot_slack below: the worst-case rounding of the accumulation, derived there -- no constant that happens to hold up to
// some m).  Candidates inside the band go to the exact pass like every other survivor.
//
// Mapping: one wavefront = 8 rows x 8 lanes; a row is one tile of consecutive candidates whose window streams through a ring in
// wave-private LDS (no workgroup barrier in the loop), a lane owns CI adjacent candidates (CI accumulators); every class
// works in chunks of 128 window points = one revolution of its ring (EDD_RING_*):
//   CI = 8 : class A, 64-candidate tiles -- full tiles and run remainders of more than 32 candidates.  Since round 5 long
//            lists whose every point is evaluated run class A on 16 candidates per lane instead (16 rows x 4 lanes, EDD16_*).
//   CI = 4 : class B, 32-candidate tiles; per lane-slot as fast as CI = 8 in practice, and only half the idle lanes on a
//            short remainder.
//   CI = 4, two tiles per row : class C, 16-candidate tiles, TWO unrelated tiles per row (lanes 0-3 / 4-7, each half row a
//            ring of its own, staged from its own tile's position): a unit of 16 such tiles takes 1.3 x a class-B unit of 8
//            (0.35 against 0.26 ms): worth it for remainders of <= 16 candidates only (kvm_ed_remainder_classes,
//            profiles/ed_class_cost.py).
// Lane stride inside a row is CI*8 B; rows are displaced by 16 B against each other ((r>>1)&1 for CI=4, r&3 for CI=8) so
// that every 16-lane group of ds_read_b128 -- which mixes lanes of four rows -- covers 16 distinct 16-B slots
// (conflict-free; SQ_LDS_BANK_CONFLICT ~ 0).  The loads are written with a clang ext_vector type: hipcc scalarises HIP's
// double2 and re-pairs it as ds_read2_b64, whose 8-B banking rules conflict 4-way here.
// The variants this file used to carry as compile-time switches -- chunk-wise staging (27 loads + 27 LDS writes per chunk),
// query values by broadcast LDS reads, a deeper register prefetch -- were measured in rounds 2-3 (HISTORY.md) and removed in
// round 5; the generated code did not change.
#include "kvm_internal.h"
#include "kvm_seqadd.h"
#include "kvm_ed_launch_model.h"
#include <hip/hip_ext.h>
bool kvm_trace_on();   // kvm_plan.hip: KVM_TRACE=1, read once

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>

typedef double ed4_d2 __attribute__((ext_vector_type(2)));

// Classes A and B stage their rows as a STREAM (round 3): a row of the wave's LDS is a ring of 128 points with an
// 80-point lookahead region behind it; every 16 window points each lane fetches ONE 16-byte piece (two points of its own
// row, four blocks ahead) and writes it where the ring has just gone dead -- 8 global loads and 13 LDS writes per 128 points
// instead of 27 + 27 per chunk (a chunk re-staged the 64 points it shares with its neighbour and restarted the register
// ring).  The arithmetic is unchanged: the same products accumulate in the same order.
// Class C (two 16-candidate tiles per row): two rings of 64 points per row.
#define EDD_RINGC_LEN 64                     // class C: ring of a half row (half a chunk: two revolutions per chunk)
#define EDD_RINGC_AHEAD 32                   // ... and its lookahead region (>= 16 + 8 window points)
// All classes take their query values from a REGISTER by DPP row broadcast instead of two broadcast LDS reads
// per sub-step: one ds_read_b64 per lane and 16 points puts q[j + (lane & 15)] into a VGPR pair, and v_fmac_f64_dpp
// row_newbcast:K (the only DPP form the DP ALU has, gfx90a+) feeds lane K's value of every 16-lane row to the FMA.  Halves the
// hot loop's LDS reads (class B, 4 candidates per lane, is LDS-bound).  The same operation: acc = fma(w, q, acc).
#define EDD_BC_CASE(K) case K: asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(q16), "v"(w)); break;
__device__ __forceinline__ void edd_fmac_bc(double &acc, double q16, double w, int k) {  // k: a constant after unrolling
    switch (k) {
        EDD_BC_CASE(0) EDD_BC_CASE(1) EDD_BC_CASE(2) EDD_BC_CASE(3) EDD_BC_CASE(4) EDD_BC_CASE(5) EDD_BC_CASE(6) EDD_BC_CASE(7)
        EDD_BC_CASE(8) EDD_BC_CASE(9) EDD_BC_CASE(10) EDD_BC_CASE(11) EDD_BC_CASE(12) EDD_BC_CASE(13) EDD_BC_CASE(14) EDD_BC_CASE(15)
    }
}
#define EDD_RING_LEN 128                     // points of the ring (= ED8_JC: one chunk is one revolution, all addresses static)
#define EDD_RING_AHEAD 80                    // lookahead region [128, 208): the start of the next revolution (>= 64 + 8 window points)
#ifndef EDD_RING_D
#define EDD_RING_D 4                         // blocks a global load is issued ahead of its LDS write
#endif
typedef double ed4_d2u __attribute__((ext_vector_type(2), aligned(8)));

// ---- class A with SIXTEEN candidates per lane (round 5; every-point evaluation only): 16 rows x 4 lanes x 16 candidates ----
// The same 64-candidate tiles, but half the bytes per FMA come back from the LDS into the VGPRs -- and those returns, not the
// LDS array, are what takes issue slots from the FMAs (profiles/r03_ed_loop_variants.log; profiles/r05_ed_a16.log: 58.3
// against 49.8 TFLOP/s for the 8-per-lane loop on one box, the library's full tiles at 47.4).
//   * lane -> (row, lane in row): lc = lane >> 4, row = (lane & 15) ^ ((lane >> 1) & 8); rows EDD16_ROW_B = 1040 bytes apart.
//     Every hardware lane group of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) then covers 16 distinct 16-byte
//     slots: slot = row + 8 lc (mod 16) = lane & 15.
//   * a row of the wave's LDS is a plain circular ring of 128 points = one chunk; no lookahead copy: a lane's read pointer is
//     a VGPR + an immediate, and each of the four lane classes steps its pointer back by the ring's length at ONE known
//     sub-step of the revolution (pair k of a revolution sits at 8 (16 lc + 20) + 16 k bytes: lc's pointer passes the end
//     at k = 54 - 8 lc, i.e. at sub-step 27 - 4 lc).
//   * stream staging: per 16-point block a lane fetches the 4 points 4 lc .. 4 lc + 3 of its row's block, two blocks ahead of
//     the LDS write; block b of the next revolution is written at sub-step 4 b (its old content died at sub-step 4 b - 2:
//     lane class 0's last read of it; lane class 3 reads the new content from sub-step 4 b + 15 on).
//   * the query values need no LDS at all: one 8-byte global load per lane and block (qc is 64 KB, every wave reads it: L2),
//     then v_fmac_f64_dpp row_newbcast as in the other classes.
//   * a window that does not end on a chunk boundary runs its last revolution to the end: qc is followed by KVM_SERIES_PAD
//     zeros, so the extra terms are x * 0 (a NaN / Inf there makes the sum NaN, which keeps the candidate for the exact pass).
#define EDD16_ROWS 16
#define EDD16_ROW_B 1040                     // bytes: 1024 of ring + one 16-byte slot of skew
#define EDD16_WAVE_B (EDD16_ROWS * EDD16_ROW_B)
#define EDD16_RINGP 16                       // register ring, pairs: sub-step G holds pairs 2G .. 2G + 9, loads 2G + 10, 2G + 11
#ifndef EDD16_AHEAD
#define EDD16_AHEAD 2                        // blocks a global load is issued ahead of its LDS write (1, 2 or 4; four: 0.299 against
#endif                                       // 0.286 ms on full tiles -- 25 accumulators and three register rings are held already)
#ifndef EDD_A16
#define EDD_A16 1                            // 0: class A on 8 lanes x 8 candidates in every mode (rounds 2 - 4)
#endif
// The 16-per-lane form computes the correlation of a PAIR of candidates (2u, 2u + 1) with three half-length sums instead of
// four (the two-parallel fast FIR structure, Karatsuba's trick on the even / odd phases of window and query):
//     acc[2u]   = EE(u) + OO(u)                      EE(u) = sum_t x[2(u + t)]     qc[2t]
//     acc[2u+1] = SS(u) - OO(u) - EE(u + 1)          OO(u) = sum_t x[2(u + t) + 1] qc[2t + 1]
//                                                    SS(u) = sum_t (x[2(u + t) + 1] + x[2(u + t) + 2]) (qc[2t] + qc[2t + 1])
// A lane's 16 candidates are 8 pairs: 9 EE + 8 OO + 8 SS = 25 accumulators, 25 FMAs per TWO window points instead of 32, and
// one addition per two points for the sums of neighbouring window values -- formed in registers: the LDS holds the plain
// window as before, a ds_read_b128 returns (x[2n], x[2n + 1]) = entry n of the even and of the odd phase.  The query's
// phases qc[2t], qc[2t + 1] and their sums come from three host-built arrays (kvm_query_dev::qk).  Algebraically the same
// sum; its rounding differs from the plain form's by a few ulps of sum |x| |qc| -- five orders of magnitude inside the guard
// band the pruner keeps around eps^2 (edd_dot_slack covers this form: its constant is derived for it), and the exact pass
// decides as before.
// profiles/r05_ed_a16k.log: the loop alone 68.2 against 57.7 TFLOP/s counted as 16 FMAs per point.
// (The plain 16-per-lane loop -- 16 accumulators, one FMA per term -- was the step before this one: profiles/r05_a16_*.log.)
// The same form serves class B on EIGHT candidates per lane (32-candidate tiles on the same 4-lane rows, 16 to a wavefront:
// 5 + 4 + 4 accumulators, 13 FMAs per two points instead of 16, and half the LDS bytes per term of the 4-per-lane class B).
template <int K>
__device__ __forceinline__ void edd_fmac_bcK(double &acc, double q16, double w) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(q16), "v"(w), "n"(K));
}
// compile-time loops: every register index and LDS offset of the 16-per-lane loop is a constant EXPRESSION (with 16
// accumulators, "constant after unrolling" left the register ring in scratch memory: 1.9 TFLOP/s)
template <int... I, class F>
__device__ __forceinline__ void edd_static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void edd_static_for(F &&f) {
    edd_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

#define EDD_ROWS 8
#define EDD_ROWLEN 192                       // staged points per row per chunk (both configurations)
#define EDD_LOADS (EDD_ROWLEN / 64)
#define EDD_RS (EDD_ROWLEN + 32)             // row stride in doubles (x8 B = multiple of 256 B), room for the row shift
#define EDD_SUB_OFF 112                      // class C: second tile of a row, in doubles (x8 B = 128 mod 256 B: the eight
                                             // lanes of a row cover the same 16-B slots as with one 32-candidate tile)
#define EDD_SUB_LEN 96                       // staged points per class-C tile per chunk
#define EDD_TAB 32                           // class C: per-tile staging table (16 row shifts, 16 starts; padded so that a
                                             // wave's region stays a multiple of 256 B), in doubles
#define EDD_WAVE_LDS (EDD_ROWS * EDD_RS + EDD_ROWLEN + EDD_TAB)
static_assert(2 * EDD_SUB_LEN == EDD_ROWLEN && ED8_JC + 8 * 8 <= EDD_ROWLEN, "chunk + tile must fit the staged (half) row");
static_assert(EDD_SUB_OFF >= EDD_SUB_LEN && EDD_SUB_OFF + EDD_SUB_LEN + 2 <= EDD_RS && (EDD_SUB_OFF * 8) % 256 == 128, "class-C half rows");
static_assert(4 * EDD_WAVE_LDS * 8 <= 65536 && EDD_WAVE_LDS % 2 == 0 && EDD_RS % 2 == 0 && (EDD_RS * 8) % 256 == 0, "LDS");
static_assert(ED8_JC % 16 == 0, "chunks are whole groups of 16 points (4 sub-steps of the register ring)");
static_assert(EDD_RINGC_LEN + EDD_RINGC_AHEAD <= EDD_SUB_LEN && 2 * EDD_RINGC_LEN == ED8_JC && EDD_RINGC_AHEAD >= 24 + 8, "class C: ring + lookahead fit a half row");
static_assert(EDD_RING_LEN == ED8_JC && EDD_RING_LEN + EDD_RING_AHEAD + 6 <= EDD_RS && EDD_RING_LEN + 16 <= EDD_ROWLEN && EDD_RING_AHEAD >= 72 + 8,
              "ring + lookahead fit a row; the query ring + its lookahead fit the query region");

template <int CI, int SUB>
struct edd_cfg {
    static constexpr int JC = ED8_JC;                    // every class: one chunk = one revolution of its ring(s)
    static constexpr bool ROW4 = CI == 16 || SUB == 4;                         // four-lane rows (EDD16_*)
    static constexpr int TILE = ROW4 ? 4 * CI : 8 * CI / SUB;                  // candidates per tile: 64 / 32 / 16
    static constexpr int TPU = ROW4 ? EDD16_ROWS : EDD_ROWS * SUB;             // tiles per unit (= per wavefront)
    static constexpr int WAVE_DOUBLES = ROW4 ? EDD16_WAVE_B / 8 : EDD_WAVE_LDS;   // a wave's LDS region
    __device__ static constexpr int row_shift(int r) { return CI == 8 ? (r & 3) * 2 : ((r >> 1) & 1) * 2; }  // doubles
};
#define EDD_TILES_A (EDD_ROWS)
#define EDD_TILES_A16 (EDD16_ROWS)
#define EDD_TILES_B16 (EDD16_ROWS)          // class B on four-lane rows (8 candidates per lane): with the 16-per-lane class A
#define EDD_TILES_B (EDD_ROWS)
#define EDD_TILES_C (EDD_ROWS * 2)

struct edd_args {
    const double *series;
    const double2 *prefix;
    kvm_query_dev qd;
    kvm_surv *surv;
    int64_t cap;
    unsigned long long *counters;
    int32_t *out_off;   // answer buffers (RSM: a wave that ends with a few survivors evaluates them exactly itself)
    double *out_dist;
    int64_t out_cap;
    double *xpart;      // device-built lists of a handful of units: the partial sums of a unit's workgroups (EDD_X_*, see edd_body)
    unsigned *xcnt;     // ... and how many of them have arrived, per unit (zero between launches)
};
// A unit of a TINY device-built list (what a selective predicate leaves: a dozen units for 2048 wave slots) is shared by
// EDD_X_WG workgroups of four waves -- 16 parts, four chunks each at m = 8192 -- instead of one: such a launch lasts as long as
// one part (a lone wave issues an fp64 instruction per ~8 cycles: 64 us for a quarter window), and the waves are there for the
// taking.  The workgroups of a unit hand their sums over through global memory; the one that arrives last finishes the unit.
#define EDD_X_WG 4
#define EDD_X_UNITS 128                                  // at most this many units in the list (x 16 parts = one wave per slot)
#define EDD_X_SLOT (8 * 64 + 64)                         // doubles per workgroup and unit: CI <= 8 sums per lane, the lanes' masks, the flag
#define EDD_X_BYTES ((size_t)EDD_X_UNITS * EDD_X_WG * EDD_X_SLOT * 8 + (size_t)EDD_X_UNITS * 4)
#define EDD_INLINE_MAX 4  // survivors per wave-unit that are evaluated in place (more: handed to kvm_exact_ed_kernel)

struct edd_cstat {
    double s1J, s2J, mc, sg;
    double e1J, e2J;     // absolute error bounds of s1J, s2J (prefix magnitudes + re-basing, kvm_prefix_err)
    kvm_stat_guard gd;   // cNSM: sigma guard, uncertainty of the mean, bound on the reference's sigma
};

// One wavefront: 8 tiles (rows) of the list `tiles`, starting at tile index t0.
// Window split: with f > 1 (f = 2 or 4) the unit is shared by f consecutive waves of the
// workgroup; wave `part` accumulates chunks [nchunks part / f, nchunks (part + 1) / f), parts > 0 hand their
// accumulators to part 0 through LDS behind one workgroup barrier, and part 0 finishes the candidates.  A unit is then
// 1/f as long, which is what lets a few thousand units drain evenly from 2048 wave slots (see the launcher).
// Every wave of a splitting workgroup reaches the barrier exactly once, whether or not it has work (`active`).
template <int CI, int SUB, bool NORM, bool ABANDON, bool CROSS = false>
__device__ __forceinline__ void edd_body(const edd_args &A, const kvm_tile *__restrict__ tiles, int64_t n_tiles, int64_t t0,
                                         double2 *lds_wave2, const int part, const int f, const bool active,
                                         const int xp = 0, const int64_t xunit = 0) {   // (CROSS: EDD_X_WG workgroups share the unit: this is number xp; the unit's index)
    constexpr int xw = CROSS ? EDD_X_WG : 1;
    using cfg = edd_cfg<CI, SUB>;
    constexpr int JC = cfg::JC;
    const int lane = threadIdx.x & 63;
    constexpr bool RING16 = CI == 16 || SUB == 4;       // 16 rows x 4 lanes x CI candidates, pairs from three sums (see EDD16_*):
    static_assert(!RING16 || (CI == 16 && SUB == 1) || (CI == 8 && SUB == 4), "<16, 1> = class A, <8, 4> = class B on four-lane rows");
    const int row = RING16 ? ((lane & 15) ^ ((lane >> 1) & 8)) : lane >> 3;
    const int sub = SUB == 2 ? ((lane >> 2) & 1) : 0;   // which of the row's tiles this lane works on
    const int lc = RING16 ? lane >> 4 : (SUB == 2 ? (lane & 3) : (lane & 7));  // lane within the tile
    const kvm_query_dev &qd = A.qd;
    const int m = qd.m;
    const double eps2 = qd.eps2;
    const double *__restrict__ series = A.series;
    const double2 *__restrict__ prefix = A.prefix;
    const double *__restrict__ qc = qd.qc;
    const double2 *__restrict__ q_pref = qd.q_pref8;   // prefix sums of the centred query at chunk granularity
    double *wlds = reinterpret_cast<double *>(lds_wave2);
    double *qlds = wlds + EDD_ROWS * EDD_RS;
    double *tab_shift = qlds + EDD_ROWLEN;                                  // class C: shift / start of the unit's 16 tiles
    int32_t *tab_start = reinterpret_cast<int32_t *>(tab_shift + 16);
    const int nchunks = (m + JC - 1) / JC;

    // class C: the start / shift of the unit's 16 tiles in the wave's LDS (a lane stages pieces of BOTH tiles of its row); the
    // other classes need no table: every lane stages pieces of its own row
    if constexpr (SUB == 2) {
        if (lane < cfg::TPU) {
            const int32_t st = (t0 + lane < n_tiles) ? tiles[t0 + lane].start : 1;
            tab_start[lane] = st;
            tab_shift[lane] = series[st - 1];
        }
    }
    // ... and this lane's own tile
    kvm_tile mine;
    mine.start = 1; mine.count = 0; mine.run_start = 1; mine.run = 0;
    if (t0 + (RING16 ? row : row * SUB + sub) < n_tiles) mine = tiles[t0 + (RING16 ? row : row * SUB + sub)];
    const double c_row = series[mine.start - 1];
    const double hist = NORM ? kvm_run_hist_of(qd, mine.run) : 0.0;  // residue the reference's running sums may carry (kvm_hist_*)
    const int c0 = CI * lc;
    const double qbar = qd.q_mean;
    const double2 qtot = q_pref[nchunks];

    // Shifted sums of points [jlo, J) of candidate a's window, re-based to the row shift; for cNSM also the window's
    // mean - c_row, sigma and sigma guard (always from the full window).
    auto stats = [&](int a, int jlo, int J) -> edd_cstat {
        edd_cstat st;
        double s1, s2, cb, pm;
        const int64_t w0 = (int64_t)mine.start - 1 + c0 + a;
        const bool full = jlo == 0 && J == m;
        const int64_t r0 = (NORM || full) ? w0 : w0 + jlo;
        const int len0 = (NORM || full) ? m : J - jlo;
        // re-based sums and their error bounds: the prefix entries (a spike earlier in the block leaves them large) and
        // the re-basing products each lose a few ulps of their own magnitude
        auto rebase = [&](int len, double &S1, double &S2, double &E1, double &E2) {
            const double dc = cb - c_row;
            const kvm_sum_err e = kvm_prefix_err(pm);
            S1 = s1 + len * dc;
            S2 = s2 + (2.0 * dc) * s1 + len * (dc * dc);
            E1 = e.e1 + 1e-15 * (fabs(s1) + len * fabs(dc));
            E2 = e.e2 + 2.0 * fabs(dc) * e.e1 + 1e-15 * (fabs(s2) + 2.0 * fabs(dc * s1) + len * (dc * dc));
        };
        kvm_window_shifted_sums(prefix, series, r0, len0, s1, s2, cb, &pm);
        double S1, S2, E1, E2;
        rebase(len0, S1, S2, E1, E2);
        st.mc = 0.0; st.sg = 1.0;
        st.gd.g = 0.0; st.gd.mean_err = 0.0; st.gd.std_hi = 0.0;
        if (NORM) {
            st.mc = S1 / m;
            const double var = S2 / m - st.mc * st.mc, mean = c_row + st.mc;
            st.sg = sqrt(var);
            st.gd.mean_err = E1 / m + kvm_hist_mean(hist, m);
            const double var_err = E2 / m + 2.0 * fabs(st.mc) * st.gd.mean_err + kvm_hist_var(hist, m);
            st.gd.g = kvm_sigma_guard(mean, st.sg, var_err / (2.0 * var) + st.gd.mean_err / st.sg);
            st.gd.std_hi = sqrt(fmax(var, 0.0) + var_err + 1e-8 * (mean * mean + fabs(var)));
            if (!full) {
                const int len = J - jlo;
                kvm_window_shifted_sums(prefix, series, w0 + jlo, len, s1, s2, cb, &pm);
                rebase(len, S1, S2, E1, E2);
            }
        }
        st.s1J = S1;
        st.s2J = S2;
        st.e1J = E1;
        st.e2J = E2;
        return st;
    };
    // `stats` of the range [jlo, J) of candidate a from loads issued earlier (kvm_issue_window_sums): RSM any range, cNSM the
    // whole window only (jlo = 0, J = m) -- the same arithmetic in the same order as `stats`
    auto stats_from = [&](const kvm_raw_sums &raw, int a, int jlo, int J) -> edd_cstat {
        edd_cstat st;
        double s1, s2, cb, pm;
        const int64_t w0 = (int64_t)mine.start - 1 + c0 + a;
        const int len0 = J - jlo;
        kvm_finish_window_sums(raw, prefix, series, w0 + jlo, len0, s1, s2, cb, &pm);
        const double dc = cb - c_row;
        const kvm_sum_err e = kvm_prefix_err(pm);
        st.s1J = s1 + len0 * dc;
        st.s2J = s2 + (2.0 * dc) * s1 + len0 * (dc * dc);
        st.e1J = e.e1 + 1e-15 * (fabs(s1) + len0 * fabs(dc));
        st.e2J = e.e2 + 2.0 * fabs(dc) * e.e1 + 1e-15 * (fabs(s2) + 2.0 * fabs(dc * s1) + len0 * (dc * dc));
        st.mc = 0.0; st.sg = 1.0;
        st.gd.g = 0.0; st.gd.mean_err = 0.0; st.gd.std_hi = 0.0;
        if (NORM) {
            st.mc = st.s1J / m;
            const double var = st.s2J / m - st.mc * st.mc, mean = c_row + st.mc;
            st.sg = sqrt(var);
            st.gd.mean_err = st.e1J / m + kvm_hist_mean(hist, m);
            const double var_err = st.e2J / m + 2.0 * fabs(st.mc) * st.gd.mean_err + kvm_hist_var(hist, m);
            st.gd.g = kvm_sigma_guard(mean, st.sg, var_err / (2.0 * var) + st.gd.mean_err / st.sg);
            st.gd.std_hi = sqrt(fmax(var, 0.0) + var_err + 1e-8 * (mean * mean + fabs(var)));
        }
        return st;
    };
    // the whole-window sums of this lane's CI candidates, all loads in flight together (one memory latency instead of two
    // per candidate: this runs once per part before the first chunk (cNSM) and once per unit behind the last)
    auto issue_full = [&](kvm_raw_sums (&raw)[4], int a0) {   // (four at a time: 64 registers of loads in flight)
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int64_t w0 = (int64_t)mine.start - 1 + ((c0 + a0 + a < mine.count) ? c0 + a0 + a : 0);
            raw[a] = kvm_issue_window_sums(prefix, series, w0, m);
        }
    };
    static_assert(CI % 4 == 0, "candidates per lane come in fours");
    // correlation-form distance over a range of J points (st, acc and the query sums qp all taken over that range) and
    // its guard band
    //
    // THE GUARD BAND.  u = 2^-53.  Everything in the two forms below is O(1) operations on a handful of sums, except
    //     acc = sum_{j < J} x'_j qc_j,          x' = x - c_row,  qc = q - mean(q),
    // which the hot loop accumulates with one rounding per FMA.  A chain of L FMAs is off by at most L u sum |x'_j qc_j| (to
    // first order in u; Higham, Accuracy and Stability, (3.3)), and sum |x' qc| <= (s sum x'^2 + sum qc^2 / s) / 2 for any scale
    // s > 0 (AM-GM) = (S2' + T2) / 2 with s = 1 (RSM) and (S2' / sg^2 + T2) sg / 2 with s = 1 / sg (cNSM, where acc enters
    // as 2 acc / sg).  What accumulates acc:
    //   * the plain loops (classes A, B, C on 8 / 4 candidates per lane): one chain of J FMAs            -> J u (S2' + T2) / 2
    //   * the pair form (16 / 8 candidates per lane): acc[2u] = EE + OO, acc[2u+1] = SS - OO - EE', each a chain of J / 2
    //     FMAs over products whose magnitudes sum to at most (S2' + T2) / 2 for EE + OO together and, the inputs of SS being
    //     sums of two neighbours ((a + b)^2 <= 2 a^2 + 2 b^2), at most (S2' + T2) for SS, plus 2 u relative for the two
    //     roundings of each SS input and three final additions                                              -> <= 0.8 J u (S2' + T2)
    //   * a window split f ways adds f - 1 additions of partial sums (each u of the running magnitude), and x' = x - c_row is
    //     itself rounded once (u |x'|): both below 4 u (S2' + T2) / 2 in all.
    // acc enters the distance as -2 acc, so the distance inherits at most 1.6 J u (S2' + T2) + 4 u (S2' + T2); the query's own
    // sums T1, T2 (host, recursive summation over the same J terms) add J u T2 and 2 |d| J u sum |qc| <= J u (T2 + J d^2).
    // edd_dot_slack = 4 (J + 4) u (S2' + T2): 2.5 x the first-order worst case (the factor also pays for the terms of second
    // order in J u, below 1e-5 of the first up to J = 2^31 / 64).  The O(1) operations that form the distance from the sums are
    // below 32 u of the magnitudes they combine (EDD_OPS_U).  At m = 8192 the band is 3.6e-12 (S2' + T2), a third of what the
    // constant 1e-11 of rounds 2 - 5 gave; that constant covers the worst case only up to m of a few 1e4, and a window built for
    // coherent rounding (0, 1, v, v, v ...: every addition of the chain rounds the same way) loses its self-match with it at
    // m = 2^22 -- tests/test_parity_gpu.py::test_ed_guard_band_holds_against_coherent_rounding, profiles/r06_guard_band_coherent_*.log;
    // ::test_ed_guard_band_scales_with_the_query_length places eps^2 on both sides of candidates at m = 2^16 and 2^18.  The
    // rounding met on ordinary data is ~sqrt(J) u: the band costs no pruning power that matters.
    constexpr double EDD_U = 1.1102230246251565e-16;   // 2^-53
    constexpr double EDD_OPS_U = 32.0 * EDD_U;
    auto edd_dot_slack = [&](int J, double s2_scaled, double t2) -> double { return (4.0 * EDD_U) * (double)(J + 4) * (fabs(s2_scaled) + fabs(t2)); };
    auto dist_form = [&](const edd_cstat &st, double acc, int J, const double2 qp, double &slack) -> double {
        if (!NORM) {
            const double d = qbar - c_row;
            const double cross = acc + d * st.s1J;
            const double qq = qp.y + (2.0 * d) * qp.x + J * (d * d);
            slack = eps2 * 1e-9 + edd_dot_slack(J, st.s2J, qp.y + J * (d * d)) +
                    EDD_OPS_U * (fabs(st.s2J) + 2.0 * fabs(acc) + 2.0 * fabs(d * st.s1J) + fabs(qq)) + st.e2J + 2.0 * fabs(d) * st.e1J;
            return st.s2J - 2.0 * cross + qq;
        } else {
            const double mc = st.mc, inv = 1.0 / st.sg;
            const double xx = st.s2J - (2.0 * mc) * st.s1J + J * (mc * mc);           // sum (x-mu)^2
            const double xz = acc + qbar * st.s1J - mc * (qp.x + J * qbar);           // sum (x-mu) zq, zq = qc + qbar
            const double zz = qp.y + (2.0 * qbar) * qp.x + J * (qbar * qbar);         // sum zq^2
            // (over a partial range the distance depends on the mean to first order: sum_J z_x - sum_J zq is not zero)
            slack = eps2 * 1e-9 + kvm_eps2_slack(eps2, m, st.gd.g, 4.0) +
                    edd_dot_slack(J, st.s2J * inv * inv, qp.y + J * (qbar * qbar)) +
                    EDD_OPS_U * (fabs(xx) * inv * inv + 2.0 * (fabs(acc) + fabs(qbar * st.s1J) + fabs(mc * qp.x)) * inv + fabs(zz)) +
                    (st.e2J + 2.0 * fabs(mc) * st.e1J) * inv * inv + 2.0 * inv * fabs(qbar) * st.e1J +
                    2.0 * inv * (fabs(st.s1J - J * mc) * inv + fabs(qp.x + J * qbar)) * st.gd.mean_err;
            return xx * inv * inv - 2.0 * xz * inv + zz;
        }
    };

    // ---- which of my CI candidates are alive (bit a of `vm`); `fm`: degenerate sigma, survives unconditionally ----
    bool work = active;  // wave-uniform
    bool handoff = false;  // wave-uniform: stopped early with <= EDD_INLINE_MAX live candidates (ABANDON only; see HAND-OVER)
    unsigned vm = 0, fm = 0;
    // (a unit split f ways, every point evaluated: only part 0 -- which finishes the candidates -- needs the (alpha, beta)
    // filter; the other parts accumulate for every candidate of their tiles and hand over.  Each part repeating the filter
    // of all its lane's candidates was 3 / 4 of that work at f = 4, on lists whose parts are short to begin with.)
    const bool need_filter = NORM && (ABANDON || part == 0);
#pragma unroll
    for (int a = 0; a < CI; a++) {
        kvm_raw_sums raw0[4];
        if constexpr (NORM) {
            if (need_filter && a % 4 == 0) issue_full(raw0, a);
        }
        if (c0 + a < mine.count) {
            bool ok = true;
            if constexpr (NORM) {
                if (need_filter) {
                    const edd_cstat st = stats_from(raw0[a % 4], a, 0, m);
                    ok = kvm_cnsm_filter_guarded(c_row + st.mc, st.sg, qd.meanQ, qd.stdQ, qd.alpha, qd.beta, st.gd);
                    if (ok && (!(st.sg == st.sg) || !(st.sg > 0.0) || kvm_guard_degenerate(st.gd.g))) fm |= 1u << a;  // exact pass decides
                }
            }
            if (ok) vm |= 1u << a;
        }
    }
    if (NORM && need_filter) {
        unsigned cntp = (unsigned)__popc(vm);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cntp += __shfl_xor(cntp, o, 64);
        if (lane == 0 && cntp && part == 0 && xp == 0) atomicAdd(&A.counters[KVM_CNT_PASS_FILTER], (unsigned long long)cntp);
        if (cntp == 0) work = false;  // nothing in this wave passed the (alpha, beta) filter
    }

    double acc[CI];
#pragma unroll
    for (int a = 0; a < CI; a++) acc[a] = 0.0;
    const int ch_lo = nchunks * (xp * f + part) / (f * xw), ch_hi = nchunks * (xp * f + part + 1) / (f * xw);
    // ---- stream staging (classes A and B): lane = 8 row + piece stages points 2 piece, 2 piece + 1 of every 16-point block of its row
    const double *gsrc = series + ((int64_t)mine.start - 1 + 2 * lc);               // + point index of the block
    double *wdst = wlds + row * EDD_RS + cfg::row_shift(row) + 2 * lc;               // + position in the row
    ed4_d2 st[EDD_RING_D], qst[2];
    auto ring_ld = [&](int j) -> ed4_d2 {   // block at window point j (unguarded: the series is padded, KVM_SERIES_PAD)
        const ed4_d2u v = *reinterpret_cast<const ed4_d2u *>(gsrc + j);
        return ed4_d2{v.x, v.y};
    };
    auto ring_wr = [&](ed4_d2 v, int pos, bool ring, bool ahead) {
        v.x -= c_row;
        v.y -= c_row;
        if (ring) *reinterpret_cast<ed4_d2 *>(wdst + pos) = v;
        if (ahead) *reinterpret_cast<ed4_d2 *>(wdst + EDD_RING_LEN + pos) = v;
    };
    auto ring_ldq = [&](int j) -> ed4_d2 {  // 64 query values from point j: lanes 0..31, two each (qc is padded like the series)
        const ed4_d2u v = *reinterpret_cast<const ed4_d2u *>(qc + j + 2 * (lane & 31));
        return ed4_d2{v.x, v.y};
    };
    auto ring_wrq = [&](ed4_d2 v, int pos, bool ahead) {   // ahead: the first 16 also behind the ring (the next revolution's start)
        if (lane < 32) *reinterpret_cast<ed4_d2 *>(qlds + pos + 2 * lane) = v;
        if (ahead && lane < 8) *reinterpret_cast<ed4_d2 *>(qlds + EDD_RING_LEN + 2 * lane) = v;
    };
    // ---- stream staging, class C: lane = 8 row + piece stages points 2 piece, 2 piece + 1 of every 16-point block of BOTH tiles
    // of its row (two streams per row, a ring of 64 points + 32 of lookahead each, in the two halves of the row)
    constexpr bool RINGC = SUB == 2;
    const double *gsrcC[2] = {series, series};
    double *wdstC[2] = {wlds, wlds};
    double shC[2] = {0.0, 0.0};
    ed4_d2 stc[2][4];
    if constexpr (RINGC) {
        asm volatile("" ::: "memory");   // (the unit's tile table was written by lanes 0..15 above)
#pragma unroll
        for (int sb = 0; sb < 2; sb++) {
            gsrcC[sb] = series + ((int64_t)tab_start[row * 2 + sb] - 1 + 2 * (lane & 7));
            shC[sb] = tab_shift[row * 2 + sb];
            wdstC[sb] = wlds + row * EDD_RS + cfg::row_shift(row) + sb * EDD_SUB_OFF + 2 * (lane & 7);
        }
    }
    auto ringc_ld = [&](int sb, int j) -> ed4_d2 {
        const ed4_d2u v = *reinterpret_cast<const ed4_d2u *>(gsrcC[sb] + j);
        return ed4_d2{v.x, v.y};
    };
    auto ringc_wr = [&](int sb, ed4_d2 v, int pos, bool ring, bool ahead) {
        v.x -= shC[sb];
        v.y -= shC[sb];
        if (ring) *reinterpret_cast<ed4_d2 *>(wdstC[sb] + pos) = v;
        if (ahead) *reinterpret_cast<ed4_d2 *>(wdstC[sb] + EDD_RINGC_LEN + pos) = v;
    };
    // ---- stream staging, 16 candidates per lane: lane (row, lc) stages points 4 lc .. 4 lc + 3 of every 16-point block of its row
    char *const rowp16 = reinterpret_cast<char *>(lds_wave2) + row * EDD16_ROW_B;
    const double *gsrc16 = series + ((int64_t)mine.start - 1 + 4 * lc);   // + window point of the block
    char *const wdst16 = rowp16 + 32 * lc;                                  // + 128 x (block of the ring)
    auto ld16 = [&](int j, ed4_d2 &u, ed4_d2 &v) {   // (unguarded: the series is padded, KVM_SERIES_PAD)
        const ed4_d2u a = *reinterpret_cast<const ed4_d2u *>(gsrc16 + j), b = *reinterpret_cast<const ed4_d2u *>(gsrc16 + j + 2);
        u = ed4_d2{a.x, a.y};
        v = ed4_d2{b.x, b.y};
    };
    auto wr16 = [&](int b, ed4_d2 u, ed4_d2 v) {
        u.x -= c_row; u.y -= c_row; v.x -= c_row; v.y -= c_row;
        *reinterpret_cast<ed4_d2 *>(wdst16 + 128 * b) = u;
        *reinterpret_cast<ed4_d2 *>(wdst16 + 128 * b + 16) = v;
    };
    ed4_d2 pend16[EDD16_AHEAD][2], R16[EDD16_RINGP];   // blocks fetched and not yet written; the window registers (carried from chunk to chunk)
    static_assert(EDD16_AHEAD == 1 || EDD16_AHEAD == 2 || EDD16_AHEAD == 4, "8 blocks per revolution: static register indices");
    // Karatsuba form: 9 + 8 + 8 accumulators; XS16[n % 16] = R16[n].y + R16[n + 1].x (sum of the neighbouring window points
    // 2n + 1, 2n + 2 of the lane); the three query phases of the 16 taps (= 32 points) in hand and of the next 16
    constexpr int NP = CI / 2;               // pairs of candidates per lane
    double kE[NP + 1], kO[NP], kS[NP], XS16[16], qe16[2], qo16[2], qs16[2];
    const int nk = kvm_qk_len(m);
    const double *__restrict__ qe_g = qd.qk, *__restrict__ qo_g = qd.qk + nk, *__restrict__ qs_g = qd.qk + 2 * (size_t)nk;
    if constexpr (RING16) {
#pragma unroll
        for (int u = 0; u < NP + 1; u++) kE[u] = 0.0;
#pragma unroll
        for (int u = 0; u < NP; u++) kO[u] = kS[u] = 0.0;
    }
    if (work && ch_lo < ch_hi) {
        if constexpr (RING16) {
            const int jb = ch_lo * JC;
            {
                ed4_d2 u[8], v[8];
#pragma unroll
                for (int b = 0; b < 8; b++) ld16(jb + 16 * b, u[b], v[b]);
                qe16[0] = qe_g[jb / 2 + (lane & 15)];
                qo16[0] = qo_g[jb / 2 + (lane & 15)];
                qs16[0] = qs_g[jb / 2 + (lane & 15)];
#pragma unroll
                for (int i = 0; i < EDD16_AHEAD; i++) ld16(jb + 128 + 16 * i, pend16[i][0], pend16[i][1]);
#pragma unroll
                for (int b = 0; b < 8; b++) wr16(b, u[b], v[b]);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < NP + 4; i++) R16[i] = *reinterpret_cast<const ed4_d2 *>(rowp16 + 8 * CI * lc + 16 * i);
#pragma unroll
            for (int i = 0; i < NP + 3; i++) XS16[i] = R16[i].y + R16[i + 1].x;
        } else if constexpr (RINGC) {
            const int jb = ch_lo * JC;
            ed4_d2 p3[2][3];
#pragma unroll
            for (int sb = 0; sb < 2; sb++)
#pragma unroll
                for (int b = 0; b < 3; b++) p3[sb][b] = ringc_ld(sb, jb + 16 * b);
            const ed4_d2 q0v = ring_ldq(jb);
#pragma unroll
            for (int sb = 0; sb < 2; sb++)
#pragma unroll
                for (int b = 0; b < 4; b++) stc[sb][b] = ringc_ld(sb, jb + 16 * (3 + b));
            qst[0] = ring_ldq(jb + 64);
#pragma unroll
            for (int sb = 0; sb < 2; sb++)
#pragma unroll
                for (int b = 0; b < 3; b++) ringc_wr(sb, p3[sb][b], 16 * b, true, false);
            ring_wrq(q0v, 0, false);
        } else {
            const int jb = ch_lo * JC;
            ed4_d2 p6[6];
#pragma unroll
            for (int b = 0; b < 6; b++) p6[b] = ring_ld(jb + 16 * b);
            const ed4_d2 q0v = ring_ldq(jb);
#pragma unroll
            for (int b = 0; b < EDD_RING_D; b++) st[b] = ring_ld(jb + 16 * (6 + b));
            qst[0] = ring_ldq(jb + 64);
#pragma unroll
            for (int b = 0; b < 6; b++) ring_wr(p6[b], 16 * b, true, false);
            ring_wrq(q0v, 0, false);
        }
    }
    const ed4_d2 *X2 = reinterpret_cast<const ed4_d2 *>(lds_wave2) +
                       (row * (EDD_RS / 2) + cfg::row_shift(row) / 2 + sub * (EDD_SUB_OFF / 2) + (CI / 2) * lc);
    for (int ch = ch_lo; work && ch < ch_hi; ch++) {
        const int j0 = ch * JC;
        const bool more = ch + 1 < ch_hi;
        const int jlim = min(JC, m - j0);
        if constexpr (RING16) {
            {
                // One revolution of the ring = one chunk of 128 points = 64 taps = 16 sub-steps of 4 taps (8 points), as
                // compile-time loops.  Pair p of the lane = its window points CI lc + 2p, + 1 = rowp16 + 8 CI lc + 16 p; sub-step S
                // holds pairs 4S .. 4S + NP + 3 in R16[p % 16] (and the sums XS16 of pairs 4S .. 4S + NP + 2) and fetches pairs
                // 4S + NP + 4 .. + 7 for the next; lane class lc's pointer passes the end of the ring at sub-step
                // (1024 - 8 CI lc - 16 (NP + 4)) / 64 = 13 - 2 lc (16 per lane) / 14 - lc (8 per lane).
                const char *P = rowp16 + 8 * CI * lc + 16 * (NP + 4);
                const int t0 = j0 / 2;
                edd_static_for<16>([&](auto Sc) {
                    constexpr int S = decltype(Sc)::value;
                    if constexpr (S % 4 == 0) {   // the query phases of the NEXT 16 taps (issued before this turn's window points)
                        constexpr int B = S / 4;
                        qe16[(B + 1) & 1] = qe_g[t0 + 16 * (B + 1) + (lane & 15)];
                        qo16[(B + 1) & 1] = qo_g[t0 + 16 * (B + 1) + (lane & 15)];
                        qs16[(B + 1) & 1] = qs_g[t0 + 16 * (B + 1) + (lane & 15)];
                    }
                    if constexpr (S % 2 == 0) {
                        constexpr int b = S / 2;
                        wr16(b, pend16[b % EDD16_AHEAD][0], pend16[b % EDD16_AHEAD][1]);
                        ld16(j0 + 128 + 16 * (b + EDD16_AHEAD), pend16[b % EDD16_AHEAD][0], pend16[b % EDD16_AHEAD][1]);
                    }
                    constexpr int W0 = (1024 - 16 * (NP + 4)) / 64, WS = 8 * CI / 64;   // class lc steps back at sub-step W0 - WS lc
                    static_assert((1024 - 16 * (NP + 4)) % 64 == 0 && (8 * CI) % 64 == 0, "whole sub-steps");
                    if constexpr (S == W0 || S == W0 - WS || S == W0 - 2 * WS || S == W0 - 3 * WS) {
                        constexpr int cls = (W0 - S) / WS;
                        P = (lc == cls) ? P - 1024 : P;
                    }
                    edd_static_for<4>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        R16[(4 * S + NP + 4 + i) % 16] = *reinterpret_cast<const ed4_d2 *>(P + 64 * S + 16 * i);
                    });
                    // fence: the reads above stay above it (memory clobber), the FMAs below it (accumulator operands)
                    if constexpr (NP == 8)
                        asm volatile("" : "+v"(kE[0]), "+v"(kE[1]), "+v"(kE[2]), "+v"(kE[3]), "+v"(kE[4]), "+v"(kE[5]), "+v"(kE[6]), "+v"(kE[7]),
                                     "+v"(kE[NP]), "+v"(kO[0]), "+v"(kO[1]), "+v"(kO[2]), "+v"(kO[3]), "+v"(kO[4 % NP]), "+v"(kO[5 % NP]),
                                     "+v"(kO[6 % NP]), "+v"(kO[7 % NP])::"memory");
                    else
                        asm volatile("" : "+v"(kE[0]), "+v"(kE[1]), "+v"(kE[2]), "+v"(kE[3]), "+v"(kE[4]), "+v"(kO[0]), "+v"(kO[1]), "+v"(kO[2]),
                                     "+v"(kO[3]), "+v"(kS[0]), "+v"(kS[1]), "+v"(kS[2]), "+v"(kS[3])::"memory");
                    const double qev = qe16[(S / 4) & 1], qov = qo16[(S / 4) & 1], qsv = qs16[(S / 4) & 1];
                    edd_static_for<4>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        constexpr int K = 4 * (S & 3) + k, T = 4 * S + k;   // the tap's lane in the query registers; the tap itself
                        edd_static_for<NP + 1>([&](auto uc) {
                            constexpr int u = decltype(uc)::value;
                            edd_fmac_bcK<K>(kE[u], qev, R16[(T + u) % 16].x);
                        });
                        edd_static_for<NP>([&](auto uc) {
                            constexpr int u = decltype(uc)::value;
                            edd_fmac_bcK<K>(kO[u], qov, R16[(T + u) % 16].y);
                        });
                        edd_static_for<NP>([&](auto uc) {
                            constexpr int u = decltype(uc)::value;
                            edd_fmac_bcK<K>(kS[u], qsv, XS16[(T + u) % 16]);
                        });
                    });
                    // the sums the next sub-step adds (the pairs fetched above have landed by now)
                    edd_static_for<4>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        XS16[(4 * S + NP + 3 + i) % 16] = R16[(4 * S + NP + 3 + i) % 16].y + R16[(4 * S + NP + 4 + i) % 16].x;
                    });
                });
                (void)jlim;
            }
        } else if constexpr (RINGC) {
            {
                // One chunk of 128 points = two revolutions of the half rows' rings = 8 blocks of 16 points; at the start the
                // half rows hold [j0, j0 + 48).  Block B's turn writes points [j0 + 16 (B + 3), + 16) of both tiles -- fetched four
                // blocks ago -- where the ring went dead a block ago (the first two blocks of a revolution also behind the ring),
                // then fetches the blocks four further on.  A block runs only if it lies wholly inside the window (the last
                // chunk of a window may be partial); what is left of a partial block is added point by point.
                constexpr int RING = 8, U = RING / 2, AH = CI / 2 + 2;
                ed4_d2 R[RING];
#pragma unroll
                for (int j = 0; j < AH; j++) R[j] = X2[j];
                const double *q16p = qlds + (lane & 15);   // (query values by DPP row broadcast, as in classes A and B)
                double qq[2];
                qq[0] = q16p[0];
                int done_blocks = 0;
                // (GUARDED = the last, partial chunk: a test per block; a full chunk runs its eight blocks as one basic block)
                auto blocks = [&](auto guarded_tag) {
                constexpr bool GUARDED = decltype(guarded_tag)::value;
#pragma unroll
                for (int B = 0; B < 8; B++) {
                    if (!GUARDED || 16 * (B + 1) <= jlim) {   // (wave-uniform)
                        done_blocks = B + 1;
                        const int wb = (B + 3) & 3;
#pragma unroll
                        for (int sb = 0; sb < 2; sb++) {
                            ringc_wr(sb, stc[sb][B % 4], 16 * wb, true, 16 * wb < EDD_RINGC_AHEAD);
                            stc[sb][B % 4] = ringc_ld(sb, j0 + 16 * (B + 3 + 4));
                        }
                        if (B == 1) {
                            ring_wrq(qst[0], 64, false);
                            qst[1] = ring_ldq(j0 + 128);
                        }
                        if (B == 5) {
                            ring_wrq(qst[1], 0, true);
                            qst[0] = ring_ldq(j0 + 192);
                        }
                        // (the 16 query values of the NEXT block; a guarded block's successor may lie past the window: its
                        // values are staged -- the query is padded -- and never used)
                        qq[(B + 1) & 1] = q16p[16 * (B + 1)];
#pragma unroll
                        for (int g = 0; g < U; g++) {
                            const int G = B * U + g, Gx = G & 15;   // (the window rings go round twice per chunk)
                            R[(2 * g + AH) % RING] = X2[2 * Gx + AH];
                            R[(2 * g + AH + 1) % RING] = X2[2 * Gx + AH + 1];
                            // fence: the reads above stay above it (memory clobber), the FMAs below it (accumulator operands)
                            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])::"memory");
#define EDD_W(k) (((k) & 1) ? R[(2 * g + (k) / 2) % RING].y : R[(2 * g + (k) / 2) % RING].x)   // w[k] = X[c0 + 4G + k]
                            const double qv = qq[B & 1];
#pragma unroll
                            for (int k = 0; k < 4; k++)
#pragma unroll
                                for (int a = 0; a < CI; a++) edd_fmac_bc(acc[a], qv, EDD_W(a + k), 4 * g + k);
#undef EDD_W
                        }
                    }
                }
                };
                if (jlim == JC) blocks(std::false_type{});
                else blocks(std::true_type{});
                if (16 * done_blocks < jlim) {   // the rest of a partial block: the blocks it reads are in place (B + 2 at most)
                    asm volatile("" ::: "memory");
                    const double *X = reinterpret_cast<const double *>(X2) + 16 * (done_blocks & 3);
                    for (int jj = 16 * done_blocks; jj < jlim; jj++) {
                        const double qv = qlds[jj];
#pragma unroll
                        for (int a = 0; a < CI; a++) acc[a] = __builtin_fma(X[jj - 16 * done_blocks + a], qv, acc[a]);
                    }
                }
            }
        } else if (jlim == JC) {
            {
                // Classes A and B.  Window registers: a ring of RING pairs, R[p % RING] = (X[c0 + 2p], X[c0 + 2p + 1]); sub-step G
                // (4 points) reads pairs 2G .. 2G + CI/2 + 1 and, before its arithmetic, issues the LDS reads of the two window
                // pairs the next sub-step adds -- into ring slots that died a sub-step earlier.  RING / 2 sub-steps advance the
                // ring by exactly RING pairs, so every register index is a compile-time constant (no rotation moves).
                // One revolution of the ring = one chunk of 128 points = 8 blocks of 16 points = 32 sub-steps, all unrolled:
                // every LDS address and every register index is a compile-time constant.  At the start the rows hold the
                // points [j0, j0 + 96); block B's turn writes points [j0 + 16 (B + 6), + 16) -- fetched four blocks ago --
                // where the ring went dead two blocks ago (and, for the blocks that start the next revolution, behind the
                // ring as well, where this revolution's last sub-steps read them), then fetches the block four further on.
                constexpr int RING = 8, U = RING / 2, AH = CI / 2 + 2;
                ed4_d2 R[RING];
#pragma unroll
                for (int j = 0; j < AH; j++) R[j] = X2[j];
                const double *q16p = qlds + (lane & 15);
                double qq[2];
                qq[0] = q16p[0];
#pragma unroll
                for (int B = 0; B < 8; B++) {
                    {
                        const int wb = (B + 6) & 7;   // ring block the write goes to
                        ring_wr(st[B % EDD_RING_D], 16 * wb, true, 16 * wb < EDD_RING_AHEAD);
                        st[B % EDD_RING_D] = ring_ld(j0 + 16 * (B + 6 + EDD_RING_D));
                        if (B == 1) {   // the second half of this revolution's query values; fetch the next revolution's first half
                            ring_wrq(qst[0], 64, false);
                            qst[1] = ring_ldq(j0 + 128);
                        }
                        if (B == 5) {   // the next revolution's first half (its first 16 values are read by this revolution's end)
                            ring_wrq(qst[1], 0, true);
                            qst[0] = ring_ldq(j0 + 192);
                        }
                    }
                    // (the 16 query values of the NEXT block, one per lane of a 16-lane row; the last block's come from the
                    // lookahead behind the query ring, written three blocks ago)
                    qq[(B + 1) & 1] = q16p[16 * (B + 1)];
#pragma unroll
                    for (int g = 0; g < U; g++) {
                        const int G = B * U + g;
                        R[(2 * g + AH) % RING] = X2[2 * G + AH];
                        R[(2 * g + AH + 1) % RING] = X2[2 * G + AH + 1];
                        if constexpr (CI == 8)
                            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),
                                         "+v"(acc[6]), "+v"(acc[7])::"memory");
                        else
                            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])::"memory");
#define EDD_W(k) (((k) & 1) ? R[(2 * g + (k) / 2) % RING].y : R[(2 * g + (k) / 2) % RING].x)
                        const double qv = qq[B & 1];
#pragma unroll
                        for (int k = 0; k < 4; k++)
#pragma unroll
                            for (int a = 0; a < CI; a++) edd_fmac_bc(acc[a], qv, EDD_W(a + k), 4 * g + k);
#undef EDD_W
                    }
                }
            }
        } else {
            {
                // classes A and B, the last, partial chunk of the window: the rows hold [j0, j0 + 96); the points up to j0 + jlim + 72 go BEHIND
                // each other in the lookahead region, so that the row is linear from the ring's start (nothing of this chunk
                // may be overwritten), and the query values [j0 + 64, j0 + 128) arrive in place
                const ed4_d2 e10 = ring_ld(j0 + 160), e11 = ring_ld(j0 + 176), e12 = ring_ld(j0 + 192);
                ring_wr(st[0], 96, true, false);
                ring_wr(st[1], 112, true, false);
                ring_wr(st[2], 0, false, true);
                ring_wr(st[3], 16, false, true);
                ring_wr(e10, 32, false, true);
                ring_wr(e11, 48, false, true);
                ring_wr(e12, 64, false, true);
                ring_wrq(qst[0], 64, false);
                asm volatile("" ::: "memory");
                const double *X = reinterpret_cast<const double *>(X2);
                for (int jj = 0; jj < jlim; jj++) {
                    const double qv = qlds[jj];
#pragma unroll
                    for (int a = 0; a < CI; a++) acc[a] = __builtin_fma(X[jj + a], qv, acc[a]);
                }
            }
        }
        if (more) {
            // early abandon (reference behaviour): the distance is a sum of non-negative terms, so a candidate whose
            // terms over ANY range of points already exceed the pruning threshold is out.  Each part tests the range
            // it has accumulated so far: after its first chunk (random candidates die here) and, unless the unit is split,
            // every 8 chunks (a split unit belongs to a short list: the launch lasts as long as its longest unit whatever
            // the others do, a wave leaves only when all its 512 candidates are dead, and each check costs more
            // than the chunk it might save).
            constexpr int EVERY = JC >= ED8_JC ? 7 : 15;   // (chunks between two tests of an unsplit unit: ~1024 points)
            if (ABANDON && (ch == ch_lo || (f == 1 && ((ch - ch_lo) & EVERY) == EVERY))) {
                const int jlo = ch_lo * JC, J = j0 + JC;
                const double2 qa = q_pref[ch + 1], qb = q_pref[ch_lo];
                const double2 qp = make_double2(qa.x - qb.x, qa.y - qb.y);
                // (a rolled loop: this runs once per 8 chunks, and unrolling its eight prefix-sum evaluations around
                // the hot loop's live registers made the compiler spill)
                if constexpr (!NORM) {
                    // RSM: the prefix-sum loads of candidate a + 1 are issued before candidate a is finished (the rolled
                    // loop otherwise pays one memory latency per candidate: with abandoning on, this check is most of
                    // what a wave does).  Unconditional, from a valid address whether or not the candidate exists.
                    auto issue = [&](int a) {
                        const int64_t w0 = (int64_t)mine.start - 1 + ((c0 + a < mine.count) ? c0 + a : 0);
                        return kvm_issue_window_sums(prefix, series, w0 + jlo, J - jlo);
                    };
                    kvm_raw_sums nxt = issue(0);
#pragma nounroll
                    for (int a = 0; a < CI; a++) {
                        const kvm_raw_sums cur = nxt;
                        nxt = issue(a + 1 < CI ? a + 1 : a);
                        double av = acc[0];
#pragma unroll
                        for (int k = 1; k < CI; k++) av = (a == k) ? acc[k] : av;
                        if (((vm >> a) & 1u) && !((fm >> a) & 1u)) {
                            double slack;
                            const edd_cstat st = stats_from(cur, a, jlo, J);
                            if (dist_form(st, av, J - jlo, qp, slack) > eps2 + slack) vm &= ~(1u << a);
                        }
                    }
                } else {
#pragma nounroll
                    for (int a = 0; a < CI; a++) {
                        double av = acc[0];
#pragma unroll
                        for (int k = 1; k < CI; k++) av = (a == k) ? acc[k] : av;
                        if (((vm >> a) & 1u) && !((fm >> a) & 1u)) {
                            double slack;
                            const edd_cstat st = stats(a, jlo, J);
                            if (dist_form(st, av, J - jlo, qp, slack) > eps2 + slack) vm &= ~(1u << a);
                        }
                    }
                }
                if (!__any(vm != 0)) {
                    if (f == 1) return;
                    break;  // split unit: the other parts wait at the barrier below
                }
                // HAND-OVER: a wave that is left with a handful of live candidates stops walking and gives them to the exact
                // evaluation as they are.  Going on costs the wave's 512 slots for those few: the unit that holds an answer
                // walked its whole share of the window on a lone wave (up to 64 of a call's 125 us at m = 8192) only to
                // confirm what the exact evaluation has to establish anyway.  Live candidates by the dozen stay in the tiled
                // form, so what this can add to the exact pass is bounded: EDD_INLINE_MAX per unit and part.
                // (Tried instead: the pruner on the first 1024 points only, everything alive there to the exact pass --
                // as fast on the bench's queries, but a window whose head matches hands over ALL its neighbours: 8 335
                // instead of 1 099 survivors and 0.55 instead of 0.41 ms on a cNSM query from a smooth stretch.)
                int alive = __popc(vm);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) alive += __shfl_xor(alive, o, 64);
                if (alive <= EDD_INLINE_MAX) {
                    handoff = true;
                    break;
                }
            }
        }
    }
    if constexpr (RING16) {
        // the pairs' three sums -> the lane's correlations (linear: a part's share of them adds up like a plain sum's)
#pragma unroll
        for (int u = 0; u < NP; u++) {
            acc[2 * u] = kE[u] + kO[u];
            acc[2 * u + 1] = kS[u] - kO[u] - kE[u + 1];
        }
    }
    if (f > 1) {
        // hand-over of the partial sums: parts 1 .. f-1 -> part 0, through the senders' own (now idle) LDS regions
        if (part > 0 && work) {
#pragma unroll
            for (int a = 0; a < CI; a++) wlds[a * 64 + lane] = acc[a];
            if (ABANDON) {
                reinterpret_cast<unsigned *>(wlds + CI * 64)[lane] = vm;  // candidates this part has seen die
                reinterpret_cast<unsigned *>(wlds + CI * 64)[64 + lane] = handoff ? 1u : 0u;   // ... and whether its sums are whole
            }
        }
        __syncthreads();
        if (part > 0 || !work) return;
        for (int p = 1; p < f; p++) {
#pragma unroll
            for (int a = 0; a < CI; a++) acc[a] += wlds[p * cfg::WAVE_DOUBLES + a * 64 + lane];
            if (ABANDON) {
                vm &= reinterpret_cast<const unsigned *>(wlds + p * cfg::WAVE_DOUBLES + CI * 64)[lane];
                handoff = handoff || reinterpret_cast<const unsigned *>(wlds + p * cfg::WAVE_DOUBLES + CI * 64)[64 + lane] != 0u;
            }
        }
    } else if (!work) {
        return;
    }
    if constexpr (CROSS) {
        // the unit's workgroups -> the one that arrives last (this wave is part 0 of its workgroup and holds that workgroup's sums).
        // Every workgroup of the unit gets here or none does: `work` is decided by the unit's tiles alone.
        double *mine_x = A.xpart + ((size_t)xunit * EDD_X_WG + xp) * EDD_X_SLOT;
#pragma unroll
        for (int a = 0; a < CI; a++) mine_x[a * 64 + lane] = acc[a];
        reinterpret_cast<unsigned *>(mine_x + 8 * 64)[lane] = vm;
        reinterpret_cast<unsigned *>(mine_x + 8 * 64)[64 + lane] = handoff ? 1u : 0u;
        __threadfence();
        unsigned arrived = 0;
        if (lane == 0) arrived = atomicAdd(&A.xcnt[xunit], 1u);
        arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived);
        if (arrived != (unsigned)(xw - 1)) return;
        __threadfence();
        if (lane == 0) A.xcnt[xunit] = 0u;   // (for the next launch)
        for (int w = 0; w < xw; w++) {
            if (w == xp) continue;
            const double *px = A.xpart + ((size_t)xunit * EDD_X_WG + w) * EDD_X_SLOT;
#pragma unroll
            for (int a = 0; a < CI; a++) acc[a] += __builtin_nontemporal_load(px + a * 64 + lane);
            vm &= __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(px + 8 * 64) + lane);
            handoff = handoff || __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(px + 8 * 64) + 64 + lane) != 0u;
        }
    }
    unsigned keepm = 0;
    kvm_raw_sums rawf[4];
    if (ABANDON && handoff) {
        keepm = vm;   // (some part stopped early: there is no whole-window sum to test -- whatever is alive goes on)
    } else {
#pragma unroll
        for (int a = 0; a < CI; a++) {
            if (a % 4 == 0) issue_full(rawf, a);
            if ((vm >> a) & 1u) {
                bool keep = (fm >> a) & 1u;
                if (!keep) {
                    double slack;
                    const edd_cstat st = stats_from(rawf[a % 4], a, 0, m);
                    keep = !(dist_form(st, acc[a], m, qtot, slack) > eps2 + slack);  // (NaN: a poisoned prefix block -> keep)
                }
                if (keep) keepm |= 1u << a;
            }
        }
    }
    if (!NORM && A.out_off) {
        // RSM: the exact evaluation of a survivor (K/QueryEngine.java:355-358: terms (x-q)*(x-q), added in order) is a
        // 40 us chain of dependent additions in one lane whatever kernel runs it.  Done here, by the wave that found
        // the survivor, it overlaps with the rest of the pruning kernel instead of following it (the separate exact
        // pass was 6 % of a phase-2 call).  Only for a few survivors per wave; more go to kvm_exact_ed_kernel.
        int ns = __popc(keepm);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ns += __shfl_xor(ns, o, 64);
        if (ns > 0 && ns <= EDD_INLINE_MAX) {
            const double *__restrict__ qx = qd.q;
            for (int it = 0; it < ns; it++) {
                const unsigned long long owners = __ballot(keepm != 0);
                const int L = __ffsll((long long)owners) - 1;                       // wave-uniform: first lane with a survivor
                const int a_me = keepm ? __ffs((int)keepm) - 1 : 0;
                const int start = __shfl(mine.start + c0 + a_me, L, 64);           // its window start (1-based)
                if (lane == L) keepm &= keepm - 1;                                   // taken
                const double *__restrict__ xs = series + ((int64_t)start - 1);
                double dist = 0.0;
                for (int j0 = 0; j0 < m; j0 += 1024) {
                    const int cnt = min(1024, m - j0);
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int j = j0 + i * 64 + lane;
                        const double xv = xs[j < m ? j : m - 1], qv = qx[j < m ? j : m - 1];
                        wlds[i * 64 + lane] = (xv - qv) * (xv - qv);               // K/QueryEngine.java:357
                    }
                    if (lane == 0) dist = ex_seq_add(wlds, cnt, dist);              // :357, strictly in order
                    // (the reference's own early abandoning, K/QueryEngine.java:356, at this loop's granularity: the terms
                    // are >= 0, so a sum beyond eps^2 stays there; NaN compares false and runs to the end, like there)
                    if (ABANDON && __shfl((int)(dist > eps2), 0, 64)) break;
                }
                if (lane == 0) {
                    atomicAdd(&A.counters[KVM_CNT_RECHECKED], 1ULL);
                    if (dist <= eps2) {
                        const unsigned long long o = atomicAdd(&A.counters[KVM_CNT_ANSWERS], 1ULL);
                        if ((int64_t)o < A.out_cap) {
                            A.out_off[o] = start;
                            A.out_dist[o] = sqrt(dist);
                        }
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int a = 0; a < CI; a++) {
        if ((keepm >> a) & 1u) {
            unsigned long long idx = atomicAdd(&A.counters[KVM_CNT_CANDS1], 1ULL);
            if ((int64_t)idx < A.cap) A.surv[idx] = kvm_surv{mine.start + c0 + a, mine.run};
        }
    }
}

// ---- tile trimming shared by the two retile kernels: a tile keeps the span of its surviving candidates (bit l of `mask` =
// lane l's candidate survives), goes to class A (33..64) or B (1..32), and is collected in LDS; one pair of global
// atomics per workgroup appends the collected tiles (one atomic per tile on a single address cost 0.3 ms at 24 000 tiles).
template <int CAP>
struct retile_out {
    kvm_tile a[CAP], b[CAP];
    int na, nb;
    unsigned long long base_a, base_b;
};
template <int CAP>
__device__ __forceinline__ void retile_collect(retile_out<CAP> &o, const kvm_tile &tl, unsigned long long mask) {
    if (mask == 0 || (threadIdx.x & 63) != 0) return;
    const int first = __ffsll((long long)mask) - 1, last = 63 - __clzll((long long)mask);
    kvm_tile out;
    out.start = tl.start + first;
    out.count = last - first + 1;
    out.run_start = tl.run_start;
    out.run = tl.run;
    // Classes as in kvm_ed_remainder_classes, except that class C is not used here: the device-built lists are what is
    // left after a selective predicate (a few hundred tiles for 2048 wave slots), the call then lasts as long as its
    // longest unit, and a class-C unit over the whole window takes 1.3 x a class-B unit (measured: the pruner of the
    // bench query with early abandoning went from 97 to 128 us with the tile that holds an answer in class C).
    if (out.count > (int)KVM_EDB_TILE) o.a[atomicAdd(&o.na, 1)] = out;
    else o.b[atomicAdd(&o.nb, 1)] = out;
}
// (all threads of the workgroup; list order is whatever the atomics give)
template <int CAP>
__device__ __forceinline__ void retile_flush(retile_out<CAP> &o, kvm_tile *__restrict__ tilesA, kvm_tile *__restrict__ tilesB,
                                             unsigned long long *__restrict__ d_counts) {
    __syncthreads();
    if (threadIdx.x == 0) {
        o.base_a = o.na ? atomicAdd(&d_counts[0], (unsigned long long)o.na) : 0;
        o.base_b = o.nb ? atomicAdd(&d_counts[1], (unsigned long long)o.nb) : 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < o.na; i += blockDim.x) tilesA[o.base_a + i] = o.a[i];
    for (int i = threadIdx.x; i < o.nb; i += blockDim.x) tilesB[o.base_b + i] = o.b[i];
}

// cNSM: trim every 64-candidate tile to the span of its candidates that pass the guarded (alpha, beta) filter and sort
// the trimmed tiles into the two classes of the pruner (33..64 -> A, 1..32 -> B); tiles without a passing candidate
// disappear.  The filter is selective in practice and its survivors are contiguous stretches of a run (mean and sigma
// move slowly with the offset), so the pruner's O(m) work shrinks by about the selectivity.  One wavefront per tile,
// one lane per candidate; 32 tiles per workgroup are collected in LDS and appended with one pair of global atomics
// (one atomic per tile on a single address cost 0.3 ms at 24 000 tiles; 16 tiles per workgroup are as fast on a selective
// filter and pay twice the atomics on a list everything passes); list order is whatever the atomics give.
#define RETILE_PER_BLOCK 32  // tiles per workgroup: 4 waves x 8 tiles (two batches of four); one pair of global atomics per workgroup
__global__ __launch_bounds__(256) void kvm_retile_cnsm_kernel(const double *__restrict__ series,
                                                              const double2 *__restrict__ prefix, kvm_query_dev qd,
                                                              const kvm_tile *__restrict__ tiles, int64_t n_tiles,
                                                              kvm_tile *__restrict__ tilesA, kvm_tile *__restrict__ tilesB,
                                                              unsigned long long *__restrict__ d_counts) {
    __shared__ retile_out<RETILE_PER_BLOCK> out;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) out.na = out.nb = 0;
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * RETILE_PER_BLOCK + wave * (RETILE_PER_BLOCK / 4);
    // Four tiles at a time: their descriptors together, then the raw prefix operands of the four windows of every lane
    // (kvm_issue_window_sums: unconditional loads from addresses valid for every lane), then the four decisions -- three memory
    // latencies per four tiles.  One tile after the other, each behind its own descriptor load, was three latencies per TILE on
    // one wave per SIMD: 31-41 us for the bench list, whatever the arithmetic behind them (profiles/retile_probe.py).
    constexpr int RB = 4;
    for (int i0 = 0; i0 < RETILE_PER_BLOCK / 4; i0 += RB) {
        if (t0 + i0 >= n_tiles) break;  // wave-uniform
        kvm_tile tls[RB];
        kvm_raw_sums raw[RB];
        double hist[RB];
#pragma unroll
        for (int u = 0; u < RB; u++) tls[u] = tiles[t0 + i0 + u < n_tiles ? t0 + i0 + u : n_tiles - 1];
#pragma unroll
        for (int u = 0; u < RB; u++) {
            raw[u] = kvm_issue_window_sums(prefix, series, (int64_t)tls[u].start - 1 + (lane < tls[u].count ? lane : 0), qd.m);
            hist[u] = kvm_run_hist_of(qd, tls[u].run);
        }
#pragma unroll
        for (int u = 0; u < RB; u++) {
            if (t0 + i0 + u >= n_tiles) break;  // wave-uniform
            const kvm_tile tl = tls[u];
            bool ok = false;
            if (lane < tl.count) {
                double s1, s2, c, pm;   // = kvm_window_mean_std_g, from the operands loaded above (same arithmetic, same order)
                kvm_finish_window_sums(raw[u], prefix, series, (int64_t)tl.start - 1 + lane, qd.m, s1, s2, c, &pm);
                const double ms = s1 / qd.m, mean = c + ms, sd = sqrt(s2 / qd.m - ms * ms);
                const kvm_stat_guard sg = kvm_make_stat_guard(c, s1, s2, qd.m, pm, hist[u]);
                ok = kvm_cnsm_filter_guarded(mean, sd, qd.meanQ, qd.stdQ, qd.alpha, qd.beta, sg);
            }
            retile_collect(out, tl, __ballot(ok));
        }
    }
    retile_flush(out, tilesA, tilesB, d_counts);
}

// RSM with the reference's early abandoning: the same trimming with the first EDR_HEAD points of the window as the
// predicate -- the sum of (x - q)^2 over any part of the window is a lower bound of the distance, and a candidate of a
// random list is dead after a hundred points.  Without it the pruner abandons per WAVE (512 candidates: one straggler
// keeps all of them computing to the next check) and a call cannot be shorter than the one unit that holds an answer
// (0.2 ms at m = 8192); with it the dead tiles never reach the pruner and the few that remain are split four ways.
// One lane per candidate, coalesced window loads, the query head by scalar loads.  `dead only if > limit`: NaN keeps.
#define EDR_HEAD 128
#define EDR_PER_BLOCK 16  // tiles per workgroup (4 per wave): almost every workgroup ends without a survivor, i.e. without
                          // atomics, so small workgroups cost nothing and put six times as many waves on a CU
__global__ __launch_bounds__(256) void kvm_retile_rsm_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                             const kvm_tile *__restrict__ tiles, int64_t n_tiles,
                                                             kvm_tile *__restrict__ tilesA, kvm_tile *__restrict__ tilesB,
                                                             unsigned long long *__restrict__ d_counts) {
    __shared__ retile_out<EDR_PER_BLOCK> out;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) out.na = out.nb = 0;
    __syncthreads();
    const int head = qd.m < EDR_HEAD ? qd.m : EDR_HEAD;
    const double lim = qd.eps2 * (1.0 + 1e-9);  // (fused, other summation order than the reference's loop: < 1e-13 relative)
    // The 64 + EDR_HEAD - 1 points a tile needs are staged once per tile in wave-private LDS (three coalesced loads per
    // lane, the next tile's in flight during the arithmetic); a lane then walks its own window there.  Reading the
    // window from global memory lane by lane cost one load latency per point: 167 us for 24 626 tiles; staged: ~15.
    __shared__ double xs[4][64 + EDR_HEAD];
    const double *__restrict__ qs = qd.q;  // uniform addresses: scalar loads, SGPR operands (no LDS traffic for the query)
    double *xw = xs[wave];
    const int64_t t0 = (int64_t)blockIdx.x * EDR_PER_BLOCK + wave * (EDR_PER_BLOCK / 4);
    auto fetch = [&](int64_t t, kvm_tile &tl, double &v0, double &v1, double &v2) {
        tl = tiles[t < n_tiles ? t : n_tiles - 1];
        const double *__restrict__ x = series + ((int64_t)tl.start - 1 + lane);  // (unguarded: the series is padded)
        v0 = x[0];
        v1 = x[64];
        v2 = x[128];
    };
    kvm_tile tl_n;
    double n0, n1, n2;
    fetch(t0, tl_n, n0, n1, n2);
    for (int i = 0; i < EDR_PER_BLOCK / 4; i++) {
        const int64_t t = t0 + i;
        if (t >= n_tiles) break;  // wave-uniform
        const kvm_tile tl = tl_n;
        xw[lane] = n0;
        xw[64 + lane] = n1;
        xw[128 + lane] = n2;
        if (i + 1 < EDR_PER_BLOCK / 4) fetch(t + 1, tl_n, n0, n1, n2);
        // blocks of 16 points: all their LDS reads and scalar loads are issued before the first use (the rolled loop
        // waited for one LDS read and one scalar load per two points)
        double a0 = 0.0, a1 = 0.0;
        int j = 0;
        for (; j + 16 <= head; j += 16) {
            double xv[16], qv[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                xv[u] = xw[lane + j + u];
                qv[u] = qs[j + u];
            }
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                const double d0 = xv[u] - qv[u], d1 = xv[u + 1] - qv[u + 1];
                a0 = __builtin_fma(d0, d0, a0);
                a1 = __builtin_fma(d1, d1, a1);
            }
            // every candidate of the tile is dead already (the sums only grow: terms >= 0, round to nearest is monotone):
            // on a random list that is the case after the first block, and the tile costs one block instead of eight
            if (!__any(lane < tl.count && !(a0 + a1 > lim))) {
                j = head;
                break;
            }
        }
        for (; j < head; j++) {
            const double d0 = xw[lane + j] - qs[j];
            a0 = __builtin_fma(d0, d0, a0);
        }
        const bool ok = lane < tl.count && !(a0 + a1 > lim);
        retile_collect(out, tl, __ballot(ok));
    }
    retile_flush(out, tilesA, tilesB, d_counts);
}

hipError_t kvm_launch_retile_rsm(const double *series, const kvm_query_dev &qd, const kvm_tile *tiles, int64_t n_tiles,
                                 kvm_tile *tilesA, kvm_tile *tilesB, unsigned long long *d_counts, hipStream_t s) {
    if (n_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(kvm_retile_rsm_kernel, dim3((unsigned)((n_tiles + EDR_PER_BLOCK - 1) / EDR_PER_BLOCK)), dim3(256),
                       0, s, series, qd, tiles, n_tiles, tilesA, tilesB, d_counts);
    return hipGetLastError();
}

hipError_t kvm_launch_retile_cnsm(const double *series, const double2 *prefix, const kvm_query_dev &qd, const kvm_tile *tiles,
                                  int64_t n_tiles, kvm_tile *tilesA, kvm_tile *tilesB, unsigned long long *d_counts,
                                  hipStream_t s) {
    if (n_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(kvm_retile_cnsm_kernel, dim3((unsigned)((n_tiles + RETILE_PER_BLOCK - 1) / RETILE_PER_BLOCK)), dim3(256),
                       0, s, series, prefix, qd, tiles, n_tiles, tilesA, tilesB, d_counts);
    return hipGetLastError();
}

// Class A on 16 candidates per lane: where every point is evaluated and the list is not tiny (edd_use_a16: decided per launch,
// `a16`; device-built lists: inside the kernel, from the real counts).  With the reference's early abandoning a wave's 1024
// candidates would have to die together before it could leave (512 with 8 per lane), and those calls are bound by their
// launch chain, not by the FMA rate: never there.
template <bool ABANDON>
__host__ __device__ constexpr bool edd_a16() { return EDD_A16 && !ABANDON; }
template <bool ABANDON>
constexpr size_t edd_lds_bytes() {
    return edd_a16<ABANDON>() ? std::max((size_t)4 * EDD16_WAVE_B, (size_t)4 * EDD_WAVE_LDS * 8) : (size_t)4 * EDD_WAVE_LDS * 8;
}

// One launch covers the three tile classes.  Workgroups [0, blocksA) run the CI=8 (or CI=16) body on `tilesA`, the next
// blocksC the two-tiles-per-row CI=4 body on `tilesC`, the rest the CI=4 body on `tilesB`; a unit = 8 (16) class-A / 8 class-B /
// 16 class-C tiles, shared by f consecutive waves of a workgroup when the window is split.
template <bool NORM, bool ABANDON>
__global__ __launch_bounds__(256, 2) void kvm_ed_dot_kernel(edd_args A, const kvm_tile *__restrict__ tilesA, int64_t nA,
                                                            const kvm_tile *__restrict__ tilesB, int64_t nB,
                                                            const kvm_tile *__restrict__ tilesC, int64_t nC, int fA, int fB,
                                                            int fC, int64_t blocksA, int64_t blocksC,
                                                            const unsigned long long *__restrict__ d_counts,
                                                            int64_t split_units, int a16, int64_t a16_min_units, int b16,
                                                            int64_t cross_units) {
    constexpr bool A16 = edd_a16<ABANDON>();
    int xw = 1;   // workgroups per unit (EDD_X_WG for a tiny device-built list)
    if (d_counts) {  // tile lists built on the device (retile kernels): the launch only knows upper bounds
        nA = (int64_t)d_counts[0];
        nB = (int64_t)d_counts[1];
        nC = (int64_t)d_counts[2];
        // the grid was sized for a 4-way split of the bounds (in 8-tile units); split only if the real lists are short
        const int64_t uA8 = (nA + EDD_TILES_A - 1) / EDD_TILES_A, uA16 = (nA + EDD_TILES_A16 - 1) / EDD_TILES_A16,
                      uB = (nB + EDD_TILES_B - 1) / EDD_TILES_B, uC = (nC + EDD_TILES_C - 1) / EDD_TILES_C;
        const bool small = uA8 + uB + uC <= split_units;
        fB = fC = small ? 4 : 1;
        // (the host's rule, on the real counts: edd_use_a16.  cNSM lists that do not fill the wave slots ran 8 per lane until
        // only part 0 of a split unit ran the (alpha, beta) filter: since then the two forms are level for one query alone and
        // the stream of eight gains 7 % with 16 per lane -- fewer FMAs to share the chip with)
        a16 = A16 && a16 && uA16 + uB + uC >= a16_min_units;
        fA = small ? 4 : ((a16 && uA16 + uB + uC <= split_units) ? 2 : 1);
        blocksA = ((a16 ? uA16 : uA8) * fA + 3) / 4;
        // (the second region of the grid: class C -- or, with class B on the four-lane rows, class B, whose 16-tile units
        // are the longer ones)
        b16 = b16 && a16;
        blocksC = b16 ? (((nB + EDD_TILES_B16 - 1) / EDD_TILES_B16) * fB + 3) / 4 : (uC * fC + 3) / 4;
        // a handful of units (8 tiles each, classes A and B on 8 / 4 per lane): EDD_X_WG workgroups to a unit (see EDD_X_*)
        // (cNSM only.  RSM-ED's device-built lists are what a 128-point head check leaves with early abandoning on: their parts
        // end at the first test or hand over, and sixteen of them cost a unit 8 us more than four -- 111 / 83 / 65 against
        // 104 / 99 / 56 us on three queries; the cNSM pruner of a selective query: 107 -> 74 us, every point 96 -> 64)
        if (NORM && small && !a16 && uC == 0 && uA8 + uB <= cross_units && (uA8 + uB) * EDD_X_WG <= (int64_t)gridDim.x) {
            xw = EDD_X_WG;
            blocksA = uA8 * EDD_X_WG;
            blocksC = 0;
        }
    }
    xw = __builtin_amdgcn_readfirstlane(xw);
    // (wave-uniform by construction, but after the 64-bit compares above the compiler keeps it in a VGPR, and a scalar select
    // on it -- `t0 = slot / fA * (a16 ? 16 : 8)` -- was MISCOMPILED in one build of this file: v_cmp into VCC followed by
    // s_cselect_b32 on a stale SCC, the s_and_b64 vcc, vcc, exec between them gone; units then started every 8 tiles and
    // covered 16.  So: an SGPR, and no select -- each form computes its own t0 from a constant.)
    a16 = __builtin_amdgcn_readfirstlane(a16);
    b16 = __builtin_amdgcn_readfirstlane(b16);
    // declared as double2 so that the window/query reads are provably 16-B aligned (ds_read_b128).  Dynamic: the 16-per-lane
    // form of class A needs 4 x 16 640 B, 1 KB more than a static array may have (edd_lds_bytes)
    extern __shared__ __attribute__((aligned(256))) double2 lds_all2[];
    const int wave = threadIdx.x >> 6;
    double2 *lds_wave2 = lds_all2 + wave * (EDD_WAVE_LDS / 2);
    // launch order = longest units first: class A (a unit over the whole window: 0.50 ms at m = 8192), then class C
    // (0.35), then class B (0.26)
    const int64_t blk = (int64_t)blockIdx.x;
    if (xw > 1) {   // (fA = fB = 4: a workgroup's four waves are the four parts of its share of the unit)
        const int64_t uA8 = (nA + EDD_TILES_A - 1) / EDD_TILES_A, uB = (nB + EDD_TILES_B - 1) / EDD_TILES_B;
        const int64_t unit = blk / xw;
        const int xp = (int)(blk % xw);
        if constexpr (NORM) {
            if (unit < uA8) edd_body<8, 1, NORM, ABANDON, true>(A, tilesA, nA, unit * EDD_TILES_A, lds_wave2, wave, 4, true, xp, unit);
            else if (unit < uA8 + uB) edd_body<4, 1, NORM, ABANDON, true>(A, tilesB, nB, (unit - uA8) * EDD_TILES_B, lds_wave2, wave, 4, true, xp, unit);
        }
        return;
    }
    if (blk < blocksA) {
        const int64_t slot = blk * 4 + wave;
        if constexpr (A16) {
            if (a16) {
                const int64_t t0 = (slot / fA) * EDD_TILES_A16;
                edd_body<16, 1, NORM, ABANDON>(A, tilesA, nA, t0, lds_all2 + wave * (EDD16_WAVE_B / 16), (int)(slot % fA), fA, t0 < nA);
                return;
            }
        }
        const int64_t t0 = (slot / fA) * EDD_TILES_A;
        edd_body<8, 1, NORM, ABANDON>(A, tilesA, nA, t0, lds_wave2, (int)(slot % fA), fA, t0 < nA);
    } else if (A16 && b16) {
        // class B on the four-lane rows as well (`b16`, only with a16: 16 tiles of <= 32 candidates per wavefront, 8 per lane,
        // pairs from three sums) in the second region, class C behind it -- the longer units first
        if constexpr (A16) {
            if (blk < blocksA + blocksC) {
                const int64_t slot = (blk - blocksA) * 4 + wave, t0 = (slot / fB) * EDD_TILES_B16;
                edd_body<8, 4, NORM, ABANDON>(A, tilesB, nB, t0, lds_all2 + wave * (EDD16_WAVE_B / 16), (int)(slot % fB), fB, t0 < nB);
            } else {
                const int64_t slot = (blk - blocksA - blocksC) * 4 + wave, t0 = (slot / fC) * EDD_TILES_C;
                if (d_counts && (blk - blocksA - blocksC) * 4 / fC * EDD_TILES_C >= nC) return;  // whole workgroup past the list
                edd_body<4, 2, NORM, ABANDON>(A, tilesC, nC, t0, lds_wave2, (int)(slot % fC), fC, t0 < nC);
            }
        }
    } else if (blk < blocksA + blocksC) {
        const int64_t slot = (blk - blocksA) * 4 + wave, t0 = (slot / fC) * EDD_TILES_C;
        edd_body<4, 2, NORM, ABANDON>(A, tilesC, nC, t0, lds_wave2, (int)(slot % fC), fC, t0 < nC);
    } else {
        const int64_t slot = (blk - blocksA - blocksC) * 4 + wave;
        const int64_t t0 = (slot / fB) * EDD_TILES_B;
        if (d_counts && (blk - blocksA - blocksC) * 4 / fB * EDD_TILES_B >= nB) return;  // whole workgroup past the list
        edd_body<4, 1, NORM, ABANDON>(A, tilesB, nB, t0, lds_wave2, (int)(slot % fB), fB, t0 < nB);
    }
}

// The launcher's model of a launch (window splits, tile classes): kvm_ed_launch_model.h (host-only, unit-tested on the CPU).
static_assert(EDM_TILES_A == EDD_TILES_A && EDM_TILES_A16 == EDD_TILES_A16 && EDM_TILES_B16 == EDD_TILES_B16 && EDM_TILES_B == EDD_TILES_B && EDM_TILES_C == EDD_TILES_C, "model and kernel agree on the unit sizes");
size_t kvm_ed_dot_x_bytes() { return EDD_X_BYTES; }
size_t kvm_ed_dot_x_counters_at() { return (size_t)EDD_X_UNITS * EDD_X_WG * EDD_X_SLOT * 8; }
int kvm_ed_dot_choose_classes(int m, int64_t tilesA, int64_t tilesB, int64_t tilesC, int num_cu, int early_abandon) {
    (void)early_abandon;   // (the question -- class C or not -- arises only on lists that do not fill the wave slots once, where
    return edd_choose_classes(m, tilesA, tilesB, tilesC, num_cu);   // class A runs in its 8-per-lane form)
}

hipError_t kvm_launch_ed_dot(const double *series, const double2 *prefix, int64_t n, const kvm_query_dev &qd,
                             const kvm_tile *tilesA, int64_t nA, const kvm_tile *tilesB, int64_t nB,
                             const kvm_tile *tilesC, int64_t nC, kvm_surv *surv,
                             int64_t cap, unsigned long long *counters, int num_cu, const unsigned long long *d_counts,
                             int32_t *out_off, double *out_dist, int64_t out_cap, hipStream_t s, hipEvent_t ev_start,
                             hipEvent_t ev_stop, int in_flight, int64_t force, int64_t *shape, double *xpart, unsigned *xcnt) {
    // `force` = KVM_OPT_ED_FORCE (kvmatch.h): parts of the launch shape imposed by the caller -- the tests drive every form of
    // the kernel over the same inputs with it; 0 = everything by the model below
    const int force_fa = (int)(force & 15), force_fb = (int)((force >> 4) & 15), force_a16 = (int)((force >> 8) & 3),
              force_b16 = (int)((force >> 10) & 3);
    const bool force_dev_a16_any = (force >> 14) & 1, force_dev_no_split = (force >> 15) & 1;
    (void)n;
    if (nA + nB + nC <= 0) return hipSuccess;
    // (the 16-per-lane forms read the query's phase arrays qd.qk: a query block built without them runs 8 per lane)
    const bool a16_ok = (qd.early_abandon ? edd_a16<true>() : edd_a16<false>()) && force_a16 != 1 && qd.qk != nullptr;
    bool a16 = false;
    const int64_t unitsA16 = (nA + EDD_TILES_A16 - 1) / EDD_TILES_A16;
    int64_t unitsA = (nA + EDD_TILES_A - 1) / EDD_TILES_A;
    const int64_t unitsB = (nB + EDD_TILES_B - 1) / EDD_TILES_B, unitsC = (nC + EDD_TILES_C - 1) / EDD_TILES_C;
    // (more than 64 KB of LDS per workgroup has to be allowed once per kernel AND PER DEVICE: the attribute belongs to the
    // function as loaded on the device that is current, and a process may hold contexts on several GPUs -- kvm_open_devices)
    {
        constexpr int MAX_DEV = 64;
        static std::once_flag lds_once[MAX_DEV];
        static hipError_t lds_rc[MAX_DEV];
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        auto allow = [] {
            hipError_t rc = hipSuccess;
            if (edd_lds_bytes<false>() > 65536) {
                rc = hipFuncSetAttribute((const void *)kvm_ed_dot_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)edd_lds_bytes<false>());
                if (rc == hipSuccess)
                    rc = hipFuncSetAttribute((const void *)kvm_ed_dot_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)edd_lds_bytes<false>());
            }
            return rc;
        };
        if (dev < 0 || dev >= MAX_DEV) {   // (no such box; still correct: set it at every launch)
            if ((e = allow()) != hipSuccess) return e;
        } else {
            std::call_once(lds_once[dev], [&] { lds_rc[dev] = allow(); });
            if (lds_rc[dev] != hipSuccess) return lds_rc[dev];
        }
    }
    const int64_t unitsS = unitsB + unitsC;  // the short units: one split factor for both classes
    // Window split (see edd_body): only worth it while the units are few compared with the 8 wave slots per CU, and
    // impossible with early abandon (waves leave early).  Pick the split pair with the smallest modelled makespan.
    int fA = 1, fB = 1;
    // Several queries in flight (kvm_run_jobs / kvm_run_device_jobs, or KVM_OPT_IN_FLIGHT of a caller with threads of its own):
    // the launches of the other contexts fill what this one leaves idle, so a split only has to cover this launch's SHARE of
    // the wave slots -- a part's prologue is then mostly lost work (the bench's stream of 8: 2.71e9 candidates/s unsplit
    // against 2.65e9 with the split that is fastest for the same list alone, profiles/r05_b8k_pairs.log)
    in_flight = std::max(1, std::min(in_flight, 64));
    const double slots_all = (double)num_cu * 8, slots = std::max(slots_all / in_flight, 64.0);
    bool b16 = false;
    int64_t split_units = 0, cross_units = 0;
    if (d_counts) {
        // device-built lists (cNSM): the real counts are unknown here and usually far below the bounds (a selective
        // filter leaves a few hundred units for 2048 wave slots): the grid covers a 4-way split of the bounds and the
        // kernel decides from the real counts (early abandon: only when the list cannot fill the chip anyway)
        split_units = -1;  // (bounds too large to size the grid for a split: the kernel then never splits)
        a16 = b16 = a16_ok;      // (the kernel decides from the real counts)
        if ((double)(unitsA + unitsS) < 16.0 * slots_all) {
            fA = fB = 4;
            // (profiles/ed_norm_split_sweep.py: a four-way split pays up to one unit per wave slot for long windows and up to
            // half of that for short ones, whose parts are mostly prologue; it was 4 x slots until late in round 2, which
            // cost up to 1.57 x at m = 1024)
            // (NOT of this launch's share with queries in flight: what passes a selective filter is a few hundred units, and
            // unsplit they last a whole window each -- the bench's cNSM stream loses 15 % with a quarter of the slots here,
            // profiles/r05_in_flight_ab.log)
            split_units = (qd.early_abandon || qd.m < 4096) ? (int64_t)(slots_all / 2) : (int64_t)slots_all;
            if (force_dev_no_split) split_units = 0;
        }
    } else if (qd.early_abandon) {
        // unit durations are unpredictable; split only when the list cannot fill the chip anyway (latency-bound)
        if ((double)(unitsA + unitsS) * 4.0 <= slots) fA = fB = 4;
        else if ((double)(unitsA + unitsS) * 2.0 <= slots) fA = fB = 2;
    } else if ((double)(unitsA + unitsS) < (in_flight >= 4 ? 2.0 : 16.0) * slots) {
        // (alone, the model decides up to 16 rounds of the wave slots -- a launch's tail is its own.  With four or more queries
        // in flight the other launches are this one's tail: from two rounds of the launch's SHARE on nothing is split -- in
        // the bench's stream every split of either class costs candidates/s, 2.71e9 unsplit against 2.68 / 2.69 with one class
        // split two ways and 2.65 with both four ways, profiles/r05_b8k_pairs.log)
        // (the model costs a few microseconds of the host's critical path: remember the last shape -- a query stream
        // over one index tends to repeat it)
        struct memo_t { int64_t a, b, c; int cu, m, fl, fa, fb; bool ok, a16, b16; };
        static thread_local memo_t memo = {-1, -1, -1, 0, 0, 0, 1, 1, false, false, false};
        if (memo.a == unitsA && memo.b == unitsB && memo.c == unitsC && memo.cu == num_cu && memo.m == qd.m && memo.fl == in_flight &&
            memo.ok == a16_ok) {
            fA = memo.fa;
            fB = memo.fb;
            a16 = memo.a16;
            b16 = memo.b16;
        } else {
            // the form of class A by the list's size (edd_use_a16), then the split pair of that form by the model.  Class B
            // moves to the four-lane rows with it where throughput counts (queries in flight): 2.4 % more candidates/s in the
            // bench's stream at every split pair, but its 16-tile units pack worse into ONE launch's tail -- the same list
            // alone 0.385 ms at its best pair against 0.372 (profiles/r05_b8k_ab.log, r05_b8k_pairs.log)
            a16 = a16_ok && edd_use_a16(qd.m, unitsA16 + unitsB + unitsC, slots_all);
            b16 = a16 && in_flight > 1;
            (void)edd_best_split(qd.m, (double)(a16 ? unitsA16 : unitsA), (double)(b16 ? (nB + EDD_TILES_B16 - 1) / EDD_TILES_B16 : unitsB),
                                 (double)unitsC, slots, fA, fB, b16 ? 2 : (a16 ? 1 : 0));
            memo = {unitsA, unitsB, unitsC, num_cu, qd.m, in_flight, fA, fB, a16_ok, a16, b16};
        }
    } else {
        a16 = b16 = a16_ok;   // many rounds of the wave slots: throughput alone
    }
    if (a16 && !d_counts) unitsA = unitsA16;
    if (force_fa) fA = force_fa;
    if (force_fb) fB = force_fb;
    if (force_a16 && !d_counts && a16_ok) {   // the form of class A on host-planned lists
        a16 = force_a16 == 2;
        b16 = a16 && in_flight > 1;
        unitsA = a16 ? unitsA16 : (nA + EDD_TILES_A - 1) / EDD_TILES_A;
    }
    if (force_b16) b16 = a16 && force_b16 == 2;   // class B on the four-lane rows (only with 16 per lane in class A)
    if (shape)   // (device-built lists: the kernel picks the forms from the real counts; what is reported is what it may use)
        *shape = (int64_t)fA | ((int64_t)fB << 4) | ((int64_t)(a16 ? 2 : 1) << 8) | ((int64_t)(b16 ? 2 : 1) << 10) | ((int64_t)in_flight << 16);
    if (kvm_trace_on())
        fprintf(stderr, "[kvm] ed_dot m %d unitsA %lld (%d per lane) unitsB %lld (class B %d per lane) unitsC %lld split %d,%d in flight %d\n", (int)qd.m,
                (long long)unitsA, a16 ? 16 : 8, (long long)((b16 && !d_counts) ? (nB + EDD_TILES_B16 - 1) / EDD_TILES_B16 : unitsB), b16 ? 8 : 4,
                (long long)unitsC, fA, fB, in_flight);
    const int fC = fB;
    const int64_t unitsB16 = (nB + EDD_TILES_B16 - 1) / EDD_TILES_B16;
    const int64_t unitsB_launch = (b16 && !d_counts) ? unitsB16 : unitsB;
    int64_t blocksA = (unitsA * fA + 3) / 4, blocksB = (unitsB_launch * fB + 3) / 4, blocksC = (unitsC * fC + 3) / 4;
    if (d_counts) {
        // device-built lists (classes A and B only): nA = nB = the number of tiles that were trimmed, and every trimmed
        // tile lands in exactly one class, so the lists together hold at most nA tiles = nA / 8 (+ one partly filled
        // unit per class) units.  The kernel lays the classes out from the real counts; the grid only has to cover that total (workgroups
        // that find nothing return at once, but each of them still has to be given its 63 KB of LDS first).
        const int64_t units_total = (nA + EDD_TILES_A - 1) / EDD_TILES_A + 3;   // (8 tiles per unit: a class-B unit's, the smaller)
        blocksA = (units_total * fA + 3) / 4 + 3;
        blocksB = blocksC = 0;
        // (a tiny list's units take EDD_X_WG workgroups each: the grid covers that for as many units as the bounds allow.  Not with
        // four or more queries in flight: the other launches fill the chip, and sixteen prologues per unit instead of four are then
        // work taken from them -- the cNSM stream of eight 9.0 - 9.2e9 against 9.3 - 9.4e9 candidates/s)
        if (xpart && xcnt && split_units > 0 && in_flight < 4) {
            cross_units = EDD_X_UNITS;
            blocksA = std::max<int64_t>(blocksA, std::min<int64_t>(units_total, EDD_X_UNITS) * EDD_X_WG);
        }
    }
    int64_t a16_min = edd_a16_min_units(qd.m, slots_all);
    if (force_dev_a16_any) a16_min = 0;   // device-built lists of any size through the 16-per-lane form
    dim3 grid((unsigned)(blocksA + blocksB + blocksC)), block(256);
    edd_args A{series, prefix, qd, surv, cap, counters, out_off, out_dist, out_cap, xpart, xcnt};
    // (ev_start / ev_stop: the kernel's own begin and end -- hipExtLaunchKernelGGL stamps the dispatch itself, so the time between
    // them holds neither the launch's dispatch latency nor a host that enqueues later than the GPU gets there)
#define LAUNCH(NORM, AB)                                                                                                          \
    do {                                                                                                                          \
        if (ev_start && ev_stop)                                                                                                  \
            hipExtLaunchKernelGGL((kvm_ed_dot_kernel<NORM, AB>), grid, block, edd_lds_bytes<AB>(), s, ev_start, ev_stop, 0, A, tilesA, nA, tilesB,  \
                                  nB, tilesC, nC, fA, fB, fC, blocksA, (b16 && !d_counts) ? blocksB : blocksC, d_counts, split_units, (int)a16, a16_min, (int)b16, cross_units); \
        else                                                                                                                      \
            hipLaunchKernelGGL((kvm_ed_dot_kernel<NORM, AB>), grid, block, edd_lds_bytes<AB>(), s, A, tilesA, nA, tilesB, nB, tilesC, nC, fA, fB,   \
                               fC, blocksA, (b16 && !d_counts) ? blocksB : blocksC, d_counts, split_units, (int)a16, a16_min, (int)b16, cross_units); \
    } while (0)
    if (qd.normalised) {
        if (qd.early_abandon) LAUNCH(true, true); else LAUNCH(true, false);
    } else {
        if (qd.early_abandon) LAUNCH(false, true); else LAUNCH(false, false);
    }
#undef LAUNCH
    return hipGetLastError();
}
