// kvm_dtw_pair.hip -- the banded DP of the DTW cascade (K/utils/DtwUtils.java:269-337) in its packed register form:
// H = 64 / L candidates per wavefront, each on L lanes x V cells of either colour, V ANY integer 1..8.
//
// Why.  The one-candidate-per-wave form (kvm_dtw.hip, dp_reg_sweep) needs V = a power of two cells per lane over 64 lanes:
// at rho = 204 (m = 4096, 5 %) that is V = 4 -- 205 of 256 slots in the band (80 % of the lanes do useful work) and, per
// macro-step, 40 cell operations among 51 instructions (the halo moves, the row minimum and the band-edge add are a
// per-LANE cost: 5 + 1 + 1 instructions whatever V is).  Two candidates per wave on 32 lanes x V = 7 fill 205 of 224 slots
// (92 %) and amortise the per-lane instructions over 14 cells instead of 8 (70 of 84 instructions are cell operations).
// rho = 409 (m = 8192) on 64 lanes x V = 7: 410 of 448 slots instead of 410 of 512.
//
// What is the same as in dp_reg_sweep (see the long comment there): band coordinate k = j - i + r, red/black in-place
// sweep over anti-diagonals with the even-k cells in E[] and the odd-k cells in O[], one macro-step = one E-update + one
// O-update, windows of data / query values sliding by one per macro-step with static register renaming, poisoned
// operands instead of per-cell masks, cell (0,0) seeded with 0.0, the band edge k = 2r+1 held at >= 1e20 by one add of a
// per-lane constant, row minima travelling with their row (high words first: one v_min3_u32 per slot), the row test
// `min_cost + cb[i+r+1] >= bsf` (:324-326) after every chunk in row order with cb[] = total - prefix of the chosen bound's
// terms.  Every in-matrix cell performs exactly min(min(x,y),z) + (a-b)*(a-b) in fp64 without contraction (:316).
//
// What is new.
//   * The seam.  The halo moves are whole-wave DPP shifts, so lane L-1 of one candidate would feed lane 0 of the next
//     (and back).  Every halo that crosses a seam is made >= 9e307 on arrival: the high word of the moved value is combined
//     with a per-lane constant by v_max_u32 (0 inside a candidate -- the identity on a non-negative double's high word --
//     and 0x7FE00000 in a candidate's first / last lane).  What leaks INTO a candidate's last lane from its neighbour's
//     cell 0 can only reach cells beyond the band edge, which no in-band cell reads.
//   * Window buffers for any V: the pairs (2h, 2h+1) of a lane's V values in sub-buffer h as before (consecutive lanes hit
//     consecutive 16-byte slots: no bank conflicts); an odd V leaves the second half of its last pair unused.
//     A buffer holds the L*V values behind the front plus ONE chunk of new values and is slid back after every chunk, so
//     the read pointers of a chunk are the same for every chunk.
//   * Candidates of a wave run out of step: each half has its own windows, chunk counter, row tests and work-queue pops,
//     all wave-uniform (scalar) state; what a half needs once per chunk (row test, slide, refill) is done by all 64 lanes
//     for one half after the other.  Only the hot block is lane-split.
//   * The second sweep with the row minima in full (a first row that cannot be told from its high word) starts at once in
//     the same half; while either half is in its second sweep the wave runs the full-minimum form of the chunk, in which a
//     first-sweep half still only looks at the high words (the high word of a minimum is the minimum of the high words).
// The run on the sequential path's candidates (exact cb[] arrays, ax.cb_exact) stays with dp_reg_sweep / the generic form.
#include "kvm_dtw_dp.h"
#include <type_traits>

// (a half's candidate / plan records are re-read per chunk rather than kept in scalar registers, and a chunk's global loads are
// issued after its blocks: the alternatives measured within run-to-run noise on 49 152 full DPs, profiles/r04_ab_pack_9.log, and
// were removed in round 6 -- these are the forms with the fewest live registers)
#ifndef KVM_DP_PACK_WG_PER_CU
#define KVM_DP_PACK_WG_PER_CU 2   // resident workgroups per CU = waves per SIMD
#endif

template <int L, int V>
struct pw_buf {
    static constexpr int H = 64 / L;                 // candidates per wavefront
    static constexpr int NBLK = 64 / V;              // blocks (of V macro-steps) per chunk
    static constexpr int CH = NBLK * V;              // macro-steps per chunk (<= 64: one lane per finished row in the row test)
    static constexpr int LIVE = L * V;               // values behind the front that lanes still read
    static constexpr int G = L + NBLK + 1;           // groups of V values per buffer (+1: the block after the chunk's last is
                                                     //   fetched and dropped)
    static constexpr int NSUB = (V + 1) / 2;
    static constexpr int SUB = 2 * G;                // doubles per sub-buffer
    static constexpr int SIZE = NSUB * SUB;          // doubles per buffer
    static constexpr int WAVE = 2 * H * SIZE + H * 64 + H * 8;  // A and B of every half, finished row minima, last-cell scratch
    __device__ static __forceinline__ int phys(int g, int q) { return (q >> 1) * SUB + 2 * g + (q & 1); }
    // the V values of a group: `at` = its first pair in sub-buffer 0, `ahead` groups further on (an immediate offset)
    __device__ static __forceinline__ void load(const double *at, int ahead, double (&out)[V]) {
#pragma unroll
        for (int h = 0; h < V / 2; h++) {
            const double2 t = *reinterpret_cast<const double2 *>(at + h * SUB + 2 * ahead);
            out[2 * h] = t.x;
            out[2 * h + 1] = t.y;
        }
        if constexpr (V & 1) out[V - 1] = at[(V / 2) * SUB + 2 * ahead];
    }
};

// lane i <- lane i-1 / lane i+1 of a non-negative double, the high word raised to `guard` on arrival (0: unchanged;
// 0x7FE00000 in the lanes a seam or the wave's end feeds: >= 9e307, never the minimum of anything).  Lanes without a source
// read zero and are always guarded.
#define PW_SEAM_HI 0x7FE00000u
__device__ __forceinline__ double pw_from_left(double v, unsigned guard) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double((int)max(hi, guard), lo);
}
__device__ __forceinline__ double pw_from_right(double v, unsigned guard) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double((int)max(hi, guard), lo);
}
__device__ __forceinline__ unsigned pw_u32_from_left(unsigned v, unsigned guard) {
    return max((unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true), guard);
}

// How a candidate is swept.  PW_FIRST: row minima as high words (one v_min3_u32 per slot); PW_FULL: row minima in full
// (two v_min_f64), for a candidate whose first row that may reach eps^2 could not be told from a high word -- it is swept
// again at once, in the same half.  The cells are the reference's in both.  A wave runs the chunk in PW_FULL while either
// of its halves needs it; a PW_FIRST half then still only looks at the high words.
enum { PW_FIRST = 0, PW_FULL = 1 };

// wave-uniform state of one candidate slot (half) of a wavefront
struct pw_half {
    int active;     // a candidate is being swept
    int mode;       // PW_FIRST / PW_FULL
    int decision;   // 0 none yet, 1 reject (ABANDON off: the sweep goes on to the end regardless)
    int Ic;         // first macro-step of the next chunk
    unsigned idx;
    kvm_cand cand;  // (kept in scalar registers: what the compiler cannot hold it parks in VGPR lanes, one instruction
    dp_plan pl;     //  per access -- a re-read from memory is a latency the wave's only partner on its SIMD cannot hide)
    double pref;    // sum of the cb terms below the next chunk's first
};
// What a half needs from global memory once per chunk, fetched BEFORE the chunk's blocks and used after them: with two
// waves per SIMD a wave that waits for memory leaves its partner alone at half the issue rate.
struct pw_pre {
    double r0, r1, r2;  // operands of the row test's cb term (dp_cb_term): x, qu, ql (query envelope) / U, L, q (data envelope)
    double xa, bq;      // the data / query value this lane adds to the windows for the next chunk
};

template <int L, int V, int CSTAR>
__device__ __forceinline__ void dp_pack_sweep(const double *__restrict__ series, const kvm_query_dev &qd, const kvm_dtw_aux &ax,
                                              const kvm_cand *__restrict__ in, const dp_plan *__restrict__ plans,
                                              const unsigned *__restrict__ list, int share_mode, int num_cu,
                                              int32_t *__restrict__ out_off, double *__restrict__ out_dist, int64_t cap,
                                              unsigned long long *__restrict__ counters, double *__restrict__ lds_all) {
    using BUF = pw_buf<L, V>;
    constexpr int H = BUF::H, NBLK = BUF::NBLK, CH = BUF::CH, LIVE = BUF::LIVE, SIZE = BUF::SIZE, SUB = BUF::SUB, NSUB = BUF::NSUB;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & (L - 1), hh = lane / L, pV = p * V;
    double *const wbase = lds_all + wave * BUF::WAVE;
    char *const lds_raw = reinterpret_cast<char *>(lds_all);
    const int m = qd.m, r = qd.rho;
    const double eps2 = qd.eps2;
    const bool abandon = qd.early_abandon != 0;
    const double *__restrict__ Bq = qd.q;
    // this kernel's share of the prologue's compact list: entries [0, n_in)
    const unsigned long long n_in = dp_pack_share(counters[KVM_CNT_DP_LIST], num_cu, share_mode, r);
    const int I0 = r >> 1, Ilast = m - 1 + I0;
    const bool edge_lane = p == r / V;  // the band-edge cell O[r] is cell CSTAR (= r % V) of this lane
    const int kfin = (r & 1) ? ((r - 1) >> 1) : (r >> 1);  // index of cell (m-1, m-1) (k = r) in O (r odd) / E (r even)
    const bool fin_lane = p == kfin / V;
    const double tol = 4.0 * ax.tau * eps2;
    // per-lane constants of the hot block
    const unsigned guardL = p == 0 ? PW_SEAM_HI : 0u, guardR = p == L - 1 ? PW_SEAM_HI : 0u;
    const double edge_pen = edge_lane ? DTW_INF : 0.0;
    double *const myA = wbase + hh * 2 * SIZE, *const myB = myA + SIZE;
    const double *const ja0 = myA + 2 * (L - p), *const jb0 = myB + 2 * (p + 1);
    double *const my_fin = wbase + 2 * H * SIZE + H * 64 + hh * 8;
    const unsigned my_strip = (unsigned)((char *)(wbase + 2 * H * SIZE + hh * 64) - lds_raw);

    // R[]: the row minima -- in full in the second-sweep form of the chunk; the first-sweep form works on their high words
    // alone and leaves the low words as they are (a first-sweep half never looks at them)
    double E[V], O[V], a[V], b[V], R[V];
#pragma unroll
    for (int c = 0; c < V; c++) {
        E[c] = DTW_INF; O[c] = DTW_INF; a[c] = DTW_POS_BIG; b[c] = DTW_NEG_BIG; R[c] = DTW_INF;
    }
    auto RH = [&](int c) -> unsigned { return (unsigned)__double2hiint(R[c]); };
    auto setRH = [&](int c, unsigned v) { R[c] = __hiloint2double((int)v, __double2loint(R[c])); };
    int mycap = -1;  // (CHECKED chunks) step of the chunk at which this lane's half computes cell (m-1, m-1); -1: not in this chunk

    // ---- the hot block: V macro-steps of every half, window values (an, bn) fetched for it ----
    // The window values of the NEXT block (an2, bn2) are fetched inside this one, pair by pair: value s of a set turns into a
    // slot of a[] / b[] at step s of its block and is replaced at step s of the next, so the registers of the PREVIOUS block's
    // pair h are free after step 2h+1 and the next block's pair h lands there with a whole block to spare -- two register
    // sets in rotation instead of three.
    auto block = [&](auto checked, auto mode_tag, const double(&an)[V], const double(&bn)[V], double(&an2)[V], double(&bn2)[V],
                     const double *ja, const double *jb, const int tb, const unsigned pr) {
        constexpr bool CHECKED = decltype(checked)::value;
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool FULLMIN = MODE == PW_FULL;
        double halo_l, halo_r, r_in;
        unsigned rh_in;
#pragma unroll
        for (int s = 0; s < V; s++) {
            // the row that left the previous lane's last slot moves on to this lane; a candidate's first lane starts a new row
            if constexpr (FULLMIN) r_in = pw_from_left(R[(V - s) % V], guardL);
            else rh_in = pw_u32_from_left(RH((V - s) % V), guardL);
            // ---- E-update (diagonal 2I - r) ----
            halo_l = pw_from_left(O[V - 1], guardL);
#pragma unroll
            for (int c = V - 1; c >= 0; c--) {
                if (c == 0) a[(V - s) % V] = an[s];  // slot of cell 0 at step s
                const double av = a[((c - s) % V + V) % V], bv = b[(c + s) % V];
                const double dd = (av - bv) * (av - bv);
                const double yy = c > 0 ? O[c - 1] : halo_l;
                E[c] = dp_min(dp_min(O[c], yy), E[c]) + dd;
            }
            if constexpr (CHECKED) {
                if ((r & 1) == 0 && tb + s == mycap && fin_lane) {  // r even: cell (m-1, m-1) is E[r/2] of this update
#pragma unroll
                    for (int c = 0; c < V; c++) my_fin[c] = E[c];
                }
            }
            // ---- O-update (diagonal 2I - r + 1) ----
            halo_r = pw_from_right(E[0], guardR);
            b[s % V] = bn[s];  // slot (V + s) % V: query value of O cell V-1, E cell V-1 of the next step
#pragma unroll
            for (int c = 0; c < V; c++) {
                const double av = a[((c - s) % V + V) % V], bv = b[(c + 1 + s) % V];
                const double dd = (av - bv) * (av - bv);
                const double xx = c < V - 1 ? E[c + 1] : halo_r;
                O[c] = dp_min(dp_min(xx, E[c]), O[c]) + dd;
            }
            if constexpr (CHECKED) {
                if ((r & 1) == 1 && tb + s == mycap && fin_lane) {
#pragma unroll
                    for (int c = 0; c < V; c++) my_fin[c] = O[c];
                }
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
                    setRH(((c - s) % V + V) % V, dp_umin3(c == 0 ? rh_in : RH(((c - s) % V + V) % V),
                                                          (unsigned)__double2hiint(E[c]), (unsigned)__double2hiint(O[c])));
                if (edge_lane) *reinterpret_cast<unsigned *>(lds_raw + pr + 4 * s) = RH(((CSTAR - s) % V + V) % V);
            }
            // ---- the next block's values that can land now ----
            if ((s & 1) == 1) {
                const int h = s >> 1;
                const double2 ta = *reinterpret_cast<const double2 *>(ja + h * SUB + 2), tb2 = *reinterpret_cast<const double2 *>(jb + h * SUB + 2);
                an2[2 * h] = ta.x; an2[2 * h + 1] = ta.y;
                bn2[2 * h] = tb2.x; bn2[2 * h + 1] = tb2.y;
            } else if (s == V - 1) {
                an2[V - 1] = ja[(V / 2) * SUB + 2];
                bn2[V - 1] = jb[(V / 2) * SUB + 2];
            }
        }
    };

    // ---- one chunk: NBLK blocks ----
    auto chunk = [&](auto checked, auto mode_tag) {
        constexpr bool FULLMIN = decltype(mode_tag)::value == PW_FULL;
        constexpr int NSETS = 2;
        double as[NSETS][V], bs[NSETS][V];
        const double *ja = ja0, *jb = jb0;
        BUF::load(ja, 0, as[0]);
        BUF::load(jb, 0, bs[0]);
        unsigned pr = my_strip;
        asm volatile("" : "+v"(pr));
        constexpr unsigned PR_STEP = V * (FULLMIN ? 8 : 4);
        int blk = 0;
#pragma unroll 1
        while (blk < NBLK) {
#pragma unroll
            for (int u = 0; u < NSETS; u++) {
                if (blk < NBLK) {
                    // (the group after the chunk's last is fetched and dropped)
                    block(checked, mode_tag, as[u], bs[u], as[(u + 1) % NSETS], bs[(u + 1) % NSETS], ja, jb, blk * V, pr);
                    blk++; ja += 2; jb += 2; pr += PR_STEP;
                }
            }
        }
    };

    pw_half S[H];
#pragma unroll
    for (int h = 0; h < H; h++) { S[h].active = 0; S[h].mode = 0; S[h].decision = 0; S[h].Ic = 0; S[h].pref = 0.0; }
    bool drained = false;
    unsigned long long pops = 0;

    // pop the work queue (wave-uniform; every entry of the list is a candidate that runs; the bound on `pops` is a guard)
    auto fetch = [&](pw_half &T) -> bool {
        if (drained || pops++ > n_in) { drained = true; return false; }
        // (a look before the pop, see dp_reg_sweep: a drained queue is not worth an atomic per wave)
        if (__hip_atomic_load(&counters[KVM_CNT_QUEUE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_in) { drained = true; return false; }
        unsigned long long popped = 0;
        if (lane == 0) popped = atomicAdd(&counters[KVM_CNT_QUEUE], 1ULL);
        const unsigned long long pos = kvm_uniform_u64(popped);
        if (pos >= n_in) { drained = true; return false; }
        const unsigned idx = list[pos];
        T.idx = idx;
        if (lane == 0) atomicAdd(&counters[KVM_CNT_PASS_KEOGH2], 1ULL);
        return true;
    };

    // (re)start the sweep of T's candidate in half h: windows of the first chunk, registers of the half's lanes
    auto start = [&](pw_half &T, const int h, const int mode_req) {
        const kvm_cand cand = in[T.idx];
        const dp_plan pl = plans[T.idx];
        const double *__restrict__ x = series + (cand.start - 1);
        const double mean = cand.mean, sd = cand.std;
        double *__restrict__ patch = ax.patch + ((size_t)(blockIdx.x * 4 + wave) * H + h) * 8 * (size_t)r;
        if (pl.which == 1) dp_build_patch(series, r, pl, cand.start, m, lane, patch);
        // prefix of the cb terms below the first index the row test needs (cb[I0 + 1]: chunk Ic covers cb[Ic+1 .. Ic+CH])
        double pref0 = 0.0;
        for (int base = 0; base <= I0; base += 64) {
            const int k = base + lane;
            pref0 += kvm_wave_sum((k <= I0 && k < m) ? dp_cb_term(x, qd, ax, patch, pl, cand.start, mean, sd, k) : 0.0);
        }
        T.pref = kvm_readlane_f64(pref0, 0);
        T.mode = mode_req;
        T.Ic = I0; T.decision = 0; T.active = 1;
        double *const bA = wbase + h * 2 * SIZE, *const bB = bA + SIZE;
        // linear position t of a buffer holds A[Ic - LIVE + t] / B[Ic - r + t]
        // zT[k] = (T[k+j] - mean) / std (K/NormQueryEngineDtw.java:564-567); RSM: (x - 0) / 1 == x exactly
        for (int t = lane; t < LIVE + CH; t += 64) {
            const int i = I0 - LIVE + t, j = I0 - r + t, at = BUF::phys(t / V, t % V);
            bA[at] = (i >= 0 && i < m) ? (x[i] - mean) / sd : DTW_POS_BIG;
            bB[at] = (j >= 0 && j < m) ? Bq[j] : DTW_NEG_BIG;
        }
        if (hh == h) {
#pragma unroll
            for (int c = 0; c < V; c++) {
                E[c] = DTW_INF;
                O[c] = DTW_INF;
                R[c] = DTW_INF;
                a[c] = c >= 1 ? myA[BUF::phys(L - p - 1, V - c)] : 0.0;  // A[I0 - pV - c]
                b[c] = myB[BUF::phys(p, c)];                               // B[I0 - r + pV + c]
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
        }
    };

    // before a chunk: issue the global loads the half needs after it (nothing here waits for them)
    auto pre = [&](const pw_half &T0, const int h, pw_pre &P) {
        pw_half T = T0; T.cand = in[T0.idx]; T.pl = plans[T0.idx];
        const double *__restrict__ x = series + (T.cand.start - 1);
        const double *__restrict__ patch = ax.patch + ((size_t)(blockIdx.x * 4 + wave) * H + h) * 8 * (size_t)r;
        const int k = min(T.Ic + lane + 1, m - 1);   // row test of lane u: needs cb term Ic + u + 1 (clamped: unused beyond m - 1)
        if (T.decision == 0) {
            if (T.pl.which == 0) {
                P.r0 = x[k]; P.r1 = qd.qu[k]; P.r2 = qd.ql[k];
            } else {
                dp_env_at(ax, patch, r, T.pl, T.cand.start + k, P.r0, P.r1);
                P.r2 = Bq[k];
            }
        }
        const int i = T.Ic + CH + lane, j = T.Ic + CH - r + LIVE + lane;
        P.xa = x[min(max(i, 0), m - 1)];
        P.bq = Bq[min(max(j, 0), m - 1)];
    };

    // after a chunk: row test of the rows it finished, then the end of the candidate or the next chunk's windows
    auto after = [&](pw_half &T, const int h, const bool chunk_full, const pw_pre &P0) {  // chunk_full: the chunk ran in PW_FULL
        T.cand = in[T.idx]; T.pl = plans[T.idx];
        pw_pre P = P0;
        pre(T, h, P);
        P.xa = (series + (T.cand.start - 1))[min(max(T.Ic + CH + lane, 0), m - 1)];
        const double mean = T.cand.mean, sd = T.cand.std;
        const int Ic = T.Ic;
        int t = 0;
        if (T.decision == 0) {
            // lane u: the row finished in macro-step Ic + u (row Ic + u - r), which needs cb[Ic + u + 1]
            const int I = Ic + lane, k = I + 1;
            const double inv = 1.0 / sd;
            const double tq = dp_term_q(P.r0, mean, inv, P.r1, P.r2), td = dp_term_d(P.r2, P.r0, P.r1, mean, inv);
            const double term = (lane < CH && k < m) ? (T.pl.which == 0 ? tq : td) : 0.0;
            const double inc = dp_scan_add(term, lane);
            const double cbv = T.pl.total - (T.pref + (inc - term));
            T.pref += kvm_readlane_f64(inc, 63);
            const bool valid = lane < CH && I >= r && I <= m - 2;
            const double *stripD = wbase + 2 * H * SIZE + h * 64;
            const unsigned *strip32 = reinterpret_cast<const unsigned *>(stripD);
            double value;
            if (T.mode == PW_FULL) t = dp_row_test(valid, stripD[lane], cbv, eps2, tol, false, value);
            else t = dp_row_test_hi(valid, chunk_full ? strip32[2 * lane + 1] : strip32[lane], cbv, eps2, tol);
        }
        if (t == 2) {
            if (T.mode != PW_FULL) {  // cannot tell from a high word: sweep again, now, with the minima in full
                if (lane == 0) atomicAdd(&counters[KVM_CNT_DP_RERUN], 1ULL);
                start(T, h, PW_FULL);
                return;
            }
            // within the tolerance of the approximate cb[]: the sequential path decides
            dp_emit_ambig(ax, counters, T.cand, lane);
            if (lane == 0) atomicAdd(&counters[KVM_CNT_PASS_KEOGH2], ~0ULL);  // (counted again when the exact run starts it)
            T.active = 0;
            return;
        }
        if (t == 1) {
            T.decision = 1;  // the reference returns here (:324-326) with a value above eps^2
            if (abandon) { T.active = 0; return; }
        }
        if (Ic + CH > Ilast) {  // the chunk held macro-step Ilast: cell (m-1, m-1) is in the half's scratch (:333-336)
            if (T.decision == 0) {
                const double v = (wbase + 2 * H * SIZE + H * 64 + h * 8)[kfin % V];
                if (lane == 0 && v <= eps2) {
                    const unsigned long long o = atomicAdd(&counters[KVM_CNT_ANSWERS], 1ULL);
                    if ((int64_t)o < cap) {
                        out_off[o] = T.cand.start;
                        out_dist[o] = sqrt(v);
                    }
                }
            }
            T.active = 0;
            return;
        }
        // next chunk: slide the LIVE values behind the front back by one chunk, add the CH values in front of them
        T.Ic = Ic + CH;
        double *const bA = wbase + h * 2 * SIZE, *const bB = bA + SIZE;
        constexpr int PASSES = (2 * L) / 64;
        double ta[NSUB][PASSES], tb2[NSUB][PASSES];
#pragma unroll
        for (int sb = 0; sb < NSUB; sb++)
#pragma unroll
            for (int ps = 0; ps < PASSES; ps++) {
                ta[sb][ps] = bA[sb * SUB + 2 * NBLK + 64 * ps + lane];
                tb2[sb][ps] = bB[sb * SUB + 2 * NBLK + 64 * ps + lane];
            }
#pragma unroll
        for (int sb = 0; sb < NSUB; sb++)
#pragma unroll
            for (int ps = 0; ps < PASSES; ps++) {
                bA[sb * SUB + 64 * ps + lane] = ta[sb][ps];
                bB[sb * SUB + 64 * ps + lane] = tb2[sb][ps];
            }
        if (lane < CH) {
            // zT[k] = (T[k+j] - mean) / std (K/NormQueryEngineDtw.java:564-567); RSM: (x - 0) / 1 == x exactly
            const int i = T.Ic + lane, j = T.Ic - r + LIVE + lane, at = BUF::phys(L + lane / V, lane % V);
            bA[at] = (i >= 0 && i < m) ? (P.xa - mean) / sd : DTW_POS_BIG;
            bB[at] = (j >= 0 && j < m) ? P.bq : DTW_NEG_BIG;
        }
    };

    for (;;) {
#pragma unroll
        for (int h = 0; h < H; h++)
            if (!S[h].active && fetch(S[h])) start(S[h], h, PW_FIRST);
        bool any = false, any_last = false;
        int wave_mode = PW_FIRST;  // (the registers need no conversion between modes: a half that is not in PW_FULL only ever
                                   //  looks at the high words of its minima, and a PW_FULL sweep starts from fresh ones)
#pragma unroll
        for (int h = 0; h < H; h++) {
            any |= S[h].active != 0;
            if (S[h].active && S[h].mode > wave_mode) wave_mode = S[h].mode;
            any_last |= S[h].active && S[h].Ic + CH > Ilast;
        }
        if (!any) break;
        pw_pre P[H];
#pragma unroll
        for (int h = 0; h < H; h++) {
            P[h].r0 = P[h].r1 = P[h].r2 = P[h].xa = P[h].bq = 0.0;
        }
        using M0 = std::integral_constant<int, PW_FIRST>;
        using M1 = std::integral_constant<int, PW_FULL>;
        if (any_last) {
            int cap_of[H];
#pragma unroll
            for (int h = 0; h < H; h++) {
                const int cp = Ilast - S[h].Ic;
                cap_of[h] = (S[h].active && cp >= 0 && cp < CH) ? cp : -1;
            }
            mycap = cap_of[0];
#pragma unroll
            for (int h = 1; h < H; h++)
                if (hh == h) mycap = cap_of[h];
            if (wave_mode == PW_FULL) chunk(std::true_type{}, M1{});
            else chunk(std::true_type{}, M0{});
        } else {
            if (wave_mode == PW_FULL) chunk(std::false_type{}, M1{});
            else chunk(std::false_type{}, M0{});
        }
#pragma unroll
        for (int h = 0; h < H; h++)
            if (S[h].active) after(S[h], h, wave_mode == PW_FULL, P[h]);
    }
}

// The kernels.  One per (L, V, CSTAR); early abandoning is a run-time flag (it only decides whether a rejected candidate
// leaves its half at once).  LDS: 4 waves x pw_buf<L, V>::WAVE doubles (L = 32, V = 7: 46.5 KiB -- three workgroups per CU).
template <int L, int V, int CSTAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KVM_DP_PACK_WG_PER_CU, KVM_DP_PACK_WG_PER_CU))) void kvm_dtw_dp_pack_kernel(
    const double *__restrict__ series, kvm_query_dev qd, kvm_dtw_aux ax, const kvm_cand *__restrict__ in,
    const dp_plan *__restrict__ plans, const unsigned *__restrict__ list, int share_mode, int num_cu, int32_t *__restrict__ out_off,
    double *__restrict__ out_dist, int64_t cap, unsigned long long *__restrict__ counters) {
    __shared__ __attribute__((aligned(16))) double lds_all[4 * pw_buf<L, V>::WAVE];
    dp_pack_sweep<L, V, CSTAR>(series, qd, ax, in, plans, list, share_mode, num_cu, out_off, out_dist, cap, counters, lds_all);
}

template <int L, int V, int CSTAR>
static hipError_t launch_pack_c(const double *series, const kvm_query_dev &qd, const kvm_dtw_aux &ax, const kvm_cand *in,
                                const dp_plan *plans, const unsigned *list, int share_mode, int64_t max_in, int32_t *out_off,
                                double *out_dist, int64_t cap, unsigned long long *counters, int num_cu, hipStream_t s) {
    constexpr int H = pw_buf<L, V>::H;
    int per_cu = (int)((160 * 1024) / (4 * pw_buf<L, V>::WAVE * sizeof(double)));
    if (per_cu > KVM_DP_PACK_WG_PER_CU) per_cu = KVM_DP_PACK_WG_PER_CU;
    int64_t blocks = (int64_t)num_cu * per_cu;
    const int64_t need = (max_in + 4 * H - 1) / (4 * H);
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL((kvm_dtw_dp_pack_kernel<L, V, CSTAR>), dim3((unsigned)blocks), dim3(256), 0, s, series, qd, ax, in, plans,
                       list, share_mode, num_cu, out_off, out_dist, cap, counters);
    return hipGetLastError();
}

template <int L, int V>
static hipError_t launch_pack(const double *series, const kvm_query_dev &qd, const kvm_dtw_aux &ax, const kvm_cand *in,
                              const dp_plan *plans, const unsigned *list, int share_mode, int64_t max_in, int32_t *out_off,
                              double *out_dist, int64_t cap, unsigned long long *counters, int num_cu, hipStream_t s) {
#define KVM_PACK_CASE(C)                                                                                                \
    case C:                                                                                                             \
        if constexpr (C < V)                                                                                            \
            return launch_pack_c<L, V, C>(series, qd, ax, in, plans, list, share_mode, max_in, out_off, out_dist, cap, counters, num_cu, s); \
        break;
    switch (qd.rho % V) {
        KVM_PACK_CASE(0) KVM_PACK_CASE(1) KVM_PACK_CASE(2) KVM_PACK_CASE(3)
        KVM_PACK_CASE(4) KVM_PACK_CASE(5) KVM_PACK_CASE(6) KVM_PACK_CASE(7)
    }
#undef KVM_PACK_CASE
    return hipErrorInvalidValue;
}

int kvm_dtw_pack_v(int rho) { return dp_pack_v(rho); }

hipError_t kvm_launch_dtw_dp_pack(const double *series, const kvm_query_dev &qd, const kvm_dtw_aux &ax, const kvm_cand *in,
                                  const void *plans_v, const unsigned *list, int share_mode, int64_t max_in, int32_t *out_off,
                                  double *out_dist, int64_t cap, unsigned long long *counters, int num_cu, hipStream_t s) {
    const dp_plan *plans = reinterpret_cast<const dp_plan *>(plans_v);
#define KVM_PACK_V(VV) \
    case VV: return launch_pack<32, VV>(series, qd, ax, in, plans, list, share_mode, max_in, out_off, out_dist, cap, counters, num_cu, s);
    if (dp_pack_l(qd.rho) == 32) {
        switch (dp_pack_v(qd.rho)) {
            KVM_PACK_V(3) KVM_PACK_V(4) KVM_PACK_V(5) KVM_PACK_V(6) KVM_PACK_V(7) KVM_PACK_V(8)
        }
    }
#define KVM_PACK_W(VV) \
    case VV: return launch_pack<64, VV>(series, qd, ax, in, plans, list, share_mode, max_in, out_off, out_dist, cap, counters, num_cu, s);
    if (dp_pack_l(qd.rho) == 64) {
        switch (dp_pack_v(qd.rho)) { KVM_PACK_W(3) KVM_PACK_W(5) KVM_PACK_W(6) KVM_PACK_W(7) }
    }
#undef KVM_PACK_W
#undef KVM_PACK_V
    return hipErrorInvalidValue;
}
