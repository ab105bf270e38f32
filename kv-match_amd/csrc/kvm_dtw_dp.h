// kvm_dtw_dp.h -- device helpers shared by the DP kernels of the DTW cascade (kvm_dtw.hip: prologue, generic and
// one-candidate-per-wave register forms; kvm_dtw_pair.hip: the paired / odd-V register form): the per-candidate plan
// record, the clipped data envelope, the terms of cb[], the row tests and the cross-lane primitives of the sweeps.
#pragma once
#include "kvm_dtw_ref.h"

#define DTW_INF 1e20  // K/utils/DtwUtils.java:24

__device__ __forceinline__ double dsq(double a, double b) { return (a - b) * (a - b); }

// ---------------------------------------------------------------------------------------------------
// Shared by both DP forms: wave-cooperative pieces of the exact decision (all 64 lanes of the wavefront take part).
enum { DP_REJECT = 0, DP_RUN = 1, DP_AMBIG = 2 };

__device__ __forceinline__ double dp_scan_add(double v, int lane) {  // inclusive, in lane order
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ double dp_sel_max(double a, double b) { return (b > a) ? b : a; }  // pure selections, as the
__device__ __forceinline__ double dp_sel_min(double a, double b) { return (b < a) ? b : a; }  // envelope is (:50-91)
__device__ __forceinline__ void dp_scan_maxmin(double &mx, double &mn, int lane) {  // inclusive running max / min
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double tx = __shfl_up(mx, o, 64), tn = __shfl_up(mn, o, 64);
        if (lane >= o) {
            mx = dp_sel_max(mx, tx);
            mn = dp_sel_min(mn, tn);
        }
    }
}

// What the prologue (kvm_dtw_prep_kernel) hands to the sweep, one record per candidate: which bound's terms form cb[]
// (0: query envelope, 1: data envelope; < 0: the candidate does not reach the DP), their total, and the geometry of the
// clipped data envelope.
struct dp_plan {
    int32_t which;
    int32_t cbeg, cend, sb;  // fetched chunk [cbeg, cend] (1-based points); sb = first point of the suffix patch
    double total;
    double pad;
};
static_assert(sizeof(dp_plan) == 32, "one 32-byte record per candidate");
#define DP_PLAN_REJECT (-1)
#define DP_PLAN_AMBIG (-2)

// Upper / lower data envelope at 1-based point p as the reference builds it over the fetched chunk:
// max / min of data[max(cbeg, p-r) .. min(cend, p+r)].  Interior points come from the whole-series envelope; within r of a
// chunk end from the wave's patch: running max/min from cbeg (PM) and towards cend (SM), 2r entries each.
__device__ __forceinline__ void dp_env_at(const kvm_dtw_aux &ax, const double *__restrict__ patch, int r, const dp_plan &pl,
                                          int32_t p, double &U, double &L) {
    if (p - r < pl.cbeg) {
        const int t = min(pl.cend, p + r) - pl.cbeg;
        U = patch[t];
        L = patch[2 * r + t];
    } else if (p + r > pl.cend) {
        const int t = p - r - pl.sb;
        U = patch[4 * r + t];
        L = patch[6 * r + t];
    } else {
        U = ax.env_u[p - 1];
        L = ax.env_l[p - 1];
    }
}

// The wave's envelope patch for one candidate: running max / min of the data from cbeg upwards (entries [0, 2r) max,
// [2r, 4r) min) and from cend downwards (entries [4r, 6r) / [6r, 8r), indexed from sb), computed only when the candidate's
// window comes within r of that end of the chunk.  2r points each: a handful of wave scans.
__device__ __forceinline__ void dp_build_patch(const double *__restrict__ series, int r, const dp_plan &pl, int32_t start, int m,
                                               int lane, double *__restrict__ patch) {
    const int32_t s1 = start, e1 = start + m - 1;
    if (!(r > 0 && (s1 - r < pl.cbeg || e1 + r > pl.cend))) return;
    if (s1 - r < pl.cbeg) {
        const int32_t last = min(pl.cend, pl.cbeg + 2 * r - 1);
        double cx = -__builtin_huge_val(), cn = __builtin_huge_val();
        for (int32_t base = pl.cbeg; base <= last; base += 64) {
            const int32_t t = base + lane;
            const double v = series[min(t, last) - 1];
            double mx = v, mn = v;  // (lanes past `last` repeat the last point: no effect on a running max / min)
            dp_scan_maxmin(mx, mn, lane);
            mx = dp_sel_max(cx, mx);
            mn = dp_sel_min(cn, mn);
            if (t <= last) {
                patch[t - pl.cbeg] = mx;
                patch[2 * r + (t - pl.cbeg)] = mn;
            }
            cx = kvm_readlane_f64(mx, 63);
            cn = kvm_readlane_f64(mn, 63);
        }
    }
    if (e1 + r > pl.cend) {
        double cx = -__builtin_huge_val(), cn = __builtin_huge_val();
        for (int32_t base = pl.cend; base >= pl.sb; base -= 64) {
            const int32_t t = base - lane;
            const double v = series[max(t, pl.sb) - 1];
            double mx = v, mn = v;
            dp_scan_maxmin(mx, mn, lane);
            mx = dp_sel_max(cx, mx);
            mn = dp_sel_min(cn, mn);
            if (t >= pl.sb) {
                patch[4 * r + (t - pl.sb)] = mx;
                patch[6 * r + (t - pl.sb)] = mn;
            }
            cx = kvm_readlane_f64(mx, 63);
            cn = kvm_readlane_f64(mn, 63);
        }
    }
    // the patch is read back by other lanes of this wavefront, through the CU's own L1: a WORKGROUP-scope fence (wait for
    // the stores; no cache action is needed inside a CU) -- an agent-scope __threadfence() writes the L2 back and cost
    // ~0.15 ms per candidate here
    __threadfence_block();
}

// The terms of the wave-parallel sums are formed with ONE reciprocal per candidate instead of a division per term: the
// sums are only compared within the tolerance tau, which carries the difference (a normalised value off by <= 3 ulp moves
// a term (w - e)^2 by <= 6u sqrt(term) |w|; |w| <= sqrt(m) wherever a term is non-zero, so a whole sum moves by
// <= 6u m sqrt(sum): the 1e-15 m / sqrt(eps^2) in kvm_dtw_aux::tau).  The divisions were 45 % of the prologue's
// instructions.  Bit-exact terms are only needed where additions are replayed one by one (kvm_dtw_ref.hip) and in the DP
// cells (fillA keeps its true division).
__device__ __forceinline__ double dp_term_q(double x, double mean, double inv, double u, double l) {
    return dref_term_q((x - mean) * inv, u, l);
}
__device__ __forceinline__ double dp_term_d(double qv, double U, double L, double mean, double inv) {
    const double uu = (U - mean) * inv, ll = (L - mean) * inv;
    double d = 0;
    if (qv > uu) d = dref_dsq(qv, uu);
    else if (qv < ll) d = dref_dsq(qv, ll);
    return d;
}

// term k (natural window position) of the bound chosen for cb[]: cb1[k] / cb2[k] of K/NormQueryEngineDtw.java:559-575
__device__ __forceinline__ double dp_cb_term(const double *__restrict__ x, const kvm_query_dev &qd, const kvm_dtw_aux &ax,
                                             const double *__restrict__ patch, const dp_plan &pl, int32_t start, double mean,
                                             double sd, int k) {
    const double inv = 1.0 / sd;
    if (pl.which == 0) return dp_term_q(x[k], mean, inv, qd.qu[k], qd.ql[k]);
    double U, L;
    dp_env_at(ax, patch, qd.rho, pl, start + k, U, L);
    return dp_term_d(qd.q[k], U, L, mean, inv);
}

// Row test over the macro-steps (register form) / rows (generic form) of one chunk: lane t holds the minimum `rv` of a
// completed row i and tests `min_cost + cb[i + r + 1] >= bsf` (K/utils/DtwUtils.java:324-326), valid iff i + r < m - 1.
//   approximate cb (cbv = total - prefix, off by <= 2 tau total <= 2 tau eps^2): a value within `tol` of eps^2 is ambiguous;
//   exact cb (second run, arrays from kvm_dtw_ref.hip): the reference's comparison itself.
// Returns 0: no row of the chunk reaches eps^2; 1: the first that does is certainly above (reject); 2: ambiguous;
// 3: exact mode, first row >= eps^2, its value in `value` (an answer iff <= eps^2, i.e. == eps^2).
__device__ __forceinline__ int dp_row_test(bool valid, double rv, double cbv, double E, double tol, bool exact, double &value) {
    const double B = rv + cbv;
    const unsigned long long bal = __ballot(valid && (exact ? (B >= E) : !(B < E - tol)));
    if (!bal) return 0;
    const double Bf = kvm_readlane_f64(B, __ffsll((long long)bal) - 1);
    value = Bf;
    if (exact) return 3;
    return (Bf > E + tol) ? 1 : 2;
}

__device__ __forceinline__ void dp_emit_ambig(const kvm_dtw_aux &ax, unsigned long long *__restrict__ counters,
                                              const kvm_cand &c, int lane) {
    if (lane == 0) ax.ambig[atomicAdd(&counters[KVM_CNT_AMBIG], 1ULL)] = c;
}

#define DTW_POS_BIG 1e10
#define DTW_NEG_BIG (-1e10)

__device__ __forceinline__ double dp_min(double a, double b) {
    // one v_min_f64; fmin() would add a canonicalising v_max_f64 per operand in IEEE mode
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double dp_from_left(double v, double fill) {   // lane i <- lane i-1; lane 0 <- fill
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dp_from_right(double v, double fill) {  // lane i <- lane i+1; lane 63 <- fill
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ unsigned dp_u32_from_left(unsigned v, unsigned fill) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned dp_umin3(unsigned a, unsigned b, unsigned c) {  // one v_min3_u32
    unsigned r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Row test of the first run (EXACT = false), whose row minima are carried as the HIGH WORDS of the cells only: every cell
// is a sum of squares, >= +0, so the order of two cells is the order of their bit patterns, and the minimum of the high
// words is the high word of the minimum -- one v_min3_u32 per slot and macro-step instead of two v_min_f64, one DPP move
// per hand-over instead of two.  The true minimum rv of a row lies in [rv_lo, rv_up] = high word with the low word all
// zeros / all ones (a relative width of 2^-20), and fl(rv + cb) is monotone in rv: a row is certainly below eps^2 when even
// rv_up + cb is (less the tolerance of the approximate cb), the first row that is not rejects the candidate when even
// rv_lo + cb is above, and is ambiguous otherwise -- the second run (EXACT) carries the minima in full.
__device__ __forceinline__ int dp_row_test_hi(bool valid, unsigned hi, double cbv, double E, double tol) {
    const double lo = __hiloint2double((int)hi, 0) + cbv, up = __hiloint2double((int)hi, -1) + cbv;
    const unsigned long long bal = __ballot(valid && !(up < E - tol));
    if (!bal) return 0;
    const double Bf = kvm_readlane_f64(lo, __ffsll((long long)bal) - 1);
    return (Bf > E + tol) ? 1 : 2;
}

// The packed register form (kvm_dtw_pair.hip) for band half-width rho: lanes per candidate L and cells per lane
// V = ceil((rho + 1) / L); L = 0: it has none.
//   rho = 128 .. 191, 256 .. 447   L = 64 (one candidate per wave): V = 3, 5, 6, 7 fills the lanes better than dp_reg_sweep's
//                                  next power of two (rho = 150: 151 of 192 slots instead of 151 of 256);
//   rho = 64 .. 127, 192 .. 255    L = 32 (two candidates per wave, long lists only: dp_pack_share);
//   rho < 64                       none: two cells per lane amortise the per-lane instructions no better than dp_reg_sweep's
//                                  one (m = 1024, rho = 51: 1.69 against 1.64 ms for 32 768 DPs, profiles/r04_dp_bench_m.log);
//   rho >= 448                     none: V = 8 on 64 lanes is dp_reg_sweep's own shape.
__host__ __device__ static inline int dp_pack_l(int rho) {
    if (rho < 64 || rho > 447) return 0;
    return (rho >= 256 || (rho >= 128 && rho <= 191)) ? 64 : 32;
}
__host__ __device__ static inline int dp_pack_v(int rho) {
    const int l = dp_pack_l(rho);
    return l ? (rho + l) / l : 0;
}

// How the candidates that reach the DP (R of them, in the prologue's compact list) are shared between the packed kernel
// and the one-candidate-per-wave kernel behind it.  With 64 lanes per candidate (rho = 128 .. 191, 256 .. 447) the packed form
// is the same shape with better filled lanes: it takes everything (m = 8192, rho = 409: 13.2 -> 10.9 ms for 8 192 full DPs,
// 1.97 -> 1.86 ms for a query's 477; rho = 150: 7.5 -> 6.6 ms for 24 576).  With two candidates per wave (the other rho in
// 64 .. 255) it executes 14 % fewer instructions per cell but runs two waves per SIMD instead of three, a wave's two DPs take
// 1.4 ms where a lone DP takes 0.6 - 0.9 ms, and a half that has run out of work idles through its partner's instructions:
// on 24 576 - 49 152 equally long synthetic DPs it was 3 - 9 % ahead on some boxes (profiles/r04_ab_pack_{7,9}.log) and 4 %
// behind on another (r04_ab_pack_11.log); on the bench's cNSM-DTW query -- row tests live, three in four DPs answers -- 15 %
// behind at 13 508 DPs and level at 50 299 (r04_dp_real_sizes.log).  No list length at which it is reliably ahead: the
// default never picks it; KVM_OPT_DP_KERNEL = 3 does (the GPU suite runs every DTW shape through it).
// Entries [0, share) are the packed kernel's.  mode = KVM_OPT_DP_KERNEL: 2 nothing packed, 3 everything packed.
__host__ __device__ static inline unsigned long long dp_pack_share(unsigned long long R, int num_cu, int mode, int rho) {
    (void)num_cu;
    const int l = dp_pack_l(rho);
    if (!l || mode == 2) return 0;
    if (mode == 3 || l == 64) return R;
    return 0;
}
