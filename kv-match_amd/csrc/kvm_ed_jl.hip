// kvm_ed_jl.hip -- correlation-form ED pruner with the LANES ALONG THE WINDOW (RSM-ED, cNSM-ED), gfx950.
//
// Same algebra and guard band as kvm_ed_dot.hip (acc = sum (x_j - c) qc_j is the only O(m) work; the exact pass
// decides every survivor), different mapping:
//   one wavefront = one unit of up to 16 consecutive candidates; a chunk is 1024 window points; lane l owns points
//   [16 l, 16 l + 16) of the chunk.  For its 16 points and the 16 candidates a lane needs x'[16 l .. 16 l + 31] and
//   qc[16 l .. 16 l + 15], ALL IN REGISTERS: 256 v_fma_f64 per chunk with static register indices and no LDS operand
//   in the loop.  Per chunk and lane: 16 ds_read_b128 (own segment + the next lane's segment as halo), 9 coalesced
//   16-B global loads of the series (staged through LDS only to turn "lane = consecutive pair" into "lane =
//   consecutive segment"), 8 coalesced 16-B loads of the query, which the host lays out pre-transposed
//   (qT[chunk][i][lane] = qc pair i of lane's segment) so that they go straight to registers.
//   LDS traffic per FMA is 0.19 B against 0.5 B in kvm_ed_dot's 8-candidates-per-lane form, the candidate tile is 16
//   instead of 64/32 (fewer idle slots on the short runs phase 1 produces), and a unit is ~5 us of work instead of
//   ~300 us, so the chip drains evenly (persistent waves, contiguous static shares of the unit list).
// The per-lane partial sums are combined once per unit by a transposing butterfly (17 adds): afterwards lane l
// holds the total of candidate l >> 2, and lanes l % 4 == 0 finish their candidate from the prefix sums.
//
// LDS layout of a staged chunk (wave-private, no barriers): 16-B slot of point pair p2 is p2 + p2/8, i.e. every
// lane segment (8 pairs) is followed by one spare slot, so that 16 lanes reading pair i of their segments hit 16
// distinct 16-B slots (stride 9 slots): conflict-free ds_read_b128, nearly conflict-free ds_write_b128.
#include "kvm_internal.h"

#include <type_traits>

typedef double jl_d2 __attribute__((ext_vector_type(2)));
typedef double jl_d2u __attribute__((ext_vector_type(2), aligned(8)));  // 16-B global load at 8-B alignment

#define JL_CI 16
#define JL_S 16
#define JL_PAIRS (JL_CHUNK / 2)                   // 512 pairs per chunk (+ 8 halo pairs)
#define JL_SLOTS 592                              // 65 segments x 9 slots = 585, rounded up
static_assert(JL_CHUNK == 64 * JL_S && JL_CI == 16 && JL_S == 16, "register tiling below is written for 16 x 16");
static_assert(JL_CHUNK + JL_CI <= KVM_SERIES_PAD, "staging reads up to one chunk + halo past the last window");

struct jl_args {
    const double *series;
    const double2 *prefix;
    kvm_query_dev qd;
    kvm_surv *surv;
    int64_t cap;
    unsigned long long *counters;
    const kvm_tile *tiles;
    int64_t n_units;
};

__device__ __forceinline__ unsigned long long jl_readfirstlane64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

template <bool NORM>
__global__ __launch_bounds__(256, 2) void kvm_ed_jl_kernel(jl_args A) {
    __shared__ __attribute__((aligned(256))) double2 lds_all[4 * JL_SLOTS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    jl_d2 *lds = reinterpret_cast<jl_d2 *>(lds_all) + wave * JL_SLOTS;
    const kvm_query_dev &qd = A.qd;
    const int m = qd.m;
    const int nch = (m + JL_CHUNK - 1) / JL_CHUNK;
    const double *__restrict__ series = A.series;
    const double2 *__restrict__ prefix = A.prefix;
    const jl_d2 *__restrict__ qT = reinterpret_cast<const jl_d2 *>(qd.qT);
    const kvm_tile *__restrict__ tiles = A.tiles;

    // contiguous share of the unit list for this wave
    const int64_t W = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
    const int64_t u0 = A.n_units * w / W, u1 = A.n_units * (w + 1) / W;
    if (u0 >= u1) return;

    // lane-constant LDS addresses (in 16-B slots)
    jl_d2 *wr = lds + (lane + (lane >> 3));  // + 72 i : pair 64 i + lane
    const jl_d2 *rd = lds + 9 * lane;        // + i + (i >> 3) : pair 8 lane + i, i < 16 (i >= 8: the next lane's segment)

    jl_d2 st[9];  // staged pairs of the chunk being prefetched
    auto load_x = [&](const double *src) {
        const jl_d2u *s2 = reinterpret_cast<const jl_d2u *>(src);
#pragma unroll
        for (int i = 0; i < 8; i++) st[i] = s2[i * 64 + lane];
        st[8] = s2[512 + (lane & 7)];  // halo: the 16 points after the chunk (all lanes load, lanes < 8 store)
    };
    auto store_x = [&](double c) {
#pragma unroll
        for (int i = 0; i < 8; i++) wr[72 * i] = jl_d2{st[i].x - c, st[i].y - c};
        if (lane < 8) wr[72 * 8] = jl_d2{st[8].x - c, st[8].y - c};
    };

    kvm_tile cur = tiles[u0];
    cur.start = __builtin_amdgcn_readfirstlane(cur.start);
    cur.count = __builtin_amdgcn_readfirstlane(cur.count);
    cur.run = __builtin_amdgcn_readfirstlane(cur.run);
    jl_d2 qs[8];
    load_x(series + ((int64_t)cur.start - 1));
#pragma unroll
    for (int i = 0; i < 8; i++) qs[i] = qT[i * 64 + lane];
    double c_cur = __builtin_bit_cast(double, jl_readfirstlane64(__builtin_bit_cast(unsigned long long, st[0].x)));  // = series[cur.start - 1]
    store_x(c_cur);

    const double qbar = qd.q_mean;
    const double2 qtot = qd.q_pref8[(m + ED8_JC - 1) / ED8_JC];
    const double eps2 = qd.eps2;
    const int mycand = lane >> 2;

    for (int64_t u = u0; u < u1; u++) {
        // the next unit's descriptor stays in VGPRs (a uniform vector load) until the end of this unit: no wait here
        const kvm_tile nxt_v = tiles[u + 1 < u1 ? u + 1 : u];
        const int64_t w0 = (int64_t)cur.start - 1 + (mycand < cur.count ? mycand : cur.count - 1);  // this lane's candidate, 0-based
        kvm_raw_sums raw = {};

        double acc[JL_CI];
#pragma unroll
        for (int a = 0; a < JL_CI; a++) acc[a] = 0.0;
        double c_nxt = c_cur;

        // one chunk; `last` is a compile-time constant in each of the two expansions, so that the next unit's
        // descriptor (nxt_v, still in flight during the first chunks) is only touched by the last one
        auto chunk = [&](int k, auto last_c) __attribute__((always_inline)) {
            constexpr bool last = decltype(last_c)::value;
            // this chunk's window segment: own 16 points + the next lane's 16
            jl_d2 X[16];
#pragma unroll
            for (int i = 0; i < 16; i++) X[i] = rd[i + (i >> 3)];
            // prefetch: the next chunk of this unit, or chunk 0 of the next unit
            load_x(last ? series + ((int64_t)nxt_v.start - 1) : series + ((int64_t)cur.start - 1 + (int64_t)(k + 1) * JL_CHUNK));
            const jl_d2 *qn = qT + (last ? 0 : (int64_t)(k + 1) * JL_PAIRS) + lane;
            if (last) raw = kvm_issue_window_sums(prefix, series, w0, m);
            // The prefetch is issued above the FMA block and consumed below it.  Both fences are empty asm statements
            // that take every accumulator as an operand and clobber memory: the loads cannot sink below the first one,
            // the FMAs cannot rise above it, and the wait for the prefetch (store_x) cannot rise above the second.
#define JL_FENCE()                                                                                                    \
    asm volatile(""                                                                                                   \
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),  \
                   "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]),             \
                   "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15])                                                        \
                 :                                                                                                    \
                 : "memory")
            JL_FENCE();
#define JL_X(k_) (((k_) & 1) ? X[(k_) >> 1].y : X[(k_) >> 1].x)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const double q0 = qs[i].x, q1 = qs[i].y;
#pragma unroll
                for (int a = 0; a < JL_CI; a++) acc[a] = __builtin_fma(JL_X(2 * i + a), q0, acc[a]);
#pragma unroll
                for (int a = 0; a < JL_CI; a++) acc[a] = __builtin_fma(JL_X(2 * i + 1 + a), q1, acc[a]);
                qs[i] = qn[i * 64];  // the same pair of the next chunk, loaded as soon as this one is dead ...
                JL_FENCE();          // ... and pinned there: a whole chunk of arithmetic covers its latency
            }
#undef JL_X
#undef JL_FENCE
            if (last) c_nxt = __builtin_bit_cast(double, jl_readfirstlane64(__builtin_bit_cast(unsigned long long, st[0].x)));
            double c_w = last ? c_nxt : c_cur;
            asm volatile("" : "+s"(c_w)::"memory");  // the subtractions of store_x depend on this: they stay below the fences
            store_x(c_w);
        };
        for (int k = 0; k + 1 < nch; k++) chunk(k, std::false_type{});
        chunk(nch - 1, std::true_type{});

        // ---- transposing butterfly: lane l ends with the total of candidate l >> 2 ----
        double r8[8], r4[4], r2[2], r1;
        {
            const bool hi = (lane & 32) != 0;
#pragma unroll
            for (int a = 0; a < 8; a++) {
                const double keep = hi ? acc[a + 8] : acc[a], give = hi ? acc[a] : acc[a + 8];
                r8[a] = keep + __shfl_xor(give, 32, 64);
            }
        }
        {
            const bool hi = (lane & 16) != 0;
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const double keep = hi ? r8[a + 4] : r8[a], give = hi ? r8[a] : r8[a + 4];
                r4[a] = keep + __shfl_xor(give, 16, 64);
            }
        }
        {
            const bool hi = (lane & 8) != 0;
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const double keep = hi ? r4[a + 2] : r4[a], give = hi ? r4[a] : r4[a + 2];
                r2[a] = keep + __shfl_xor(give, 8, 64);
            }
        }
        {
            const bool hi = (lane & 4) != 0;
            const double keep = hi ? r2[1] : r2[0], give = hi ? r2[0] : r2[1];
            r1 = keep + __shfl_xor(give, 4, 64);
        }
        r1 += __shfl_xor(r1, 2, 64);
        r1 += __shfl_xor(r1, 1, 64);

        // ---- finish candidate `mycand` (one lane of each group of four) ----
        bool alive = (lane & 3) == 0 && mycand < cur.count;
        bool passed = false, keep = false;
        if (alive) {
            double s1, s2, cb, pmag;
            kvm_finish_window_sums(raw, prefix, series, w0, m, s1, s2, cb, &pmag);
            const double dc = cb - c_cur;
            const double S1 = s1 + m * dc, S2 = s2 + (2.0 * dc) * s1 + m * (dc * dc);  // sums of (x - c_cur), (x - c_cur)^2
            // error bounds of S1, S2: prefix magnitudes + re-basing (same rule as kvm_ed_dot.hip)
            const kvm_sum_err pe = kvm_prefix_err(pmag);
            const double E1 = pe.e1 + 1e-15 * (fabs(s1) + m * fabs(dc));
            const double E2 = pe.e2 + 2.0 * fabs(dc) * pe.e1 + 1e-15 * (fabs(s2) + 2.0 * fabs(dc * s1) + m * (dc * dc));
            if (!NORM) {
                const double d = qbar - c_cur;
                const double cross = r1 + d * S1;
                const double qq = qtot.y + (2.0 * d) * qtot.x + m * (d * d);
                const double slack = eps2 * 1e-9 + 1e-11 * (fabs(S2) + 2.0 * fabs(r1) + 2.0 * fabs(d * S1) + fabs(qq)) +
                                     E2 + 2.0 * fabs(d) * E1;
                keep = !((S2 - 2.0 * cross + qq) > eps2 + slack);  // (NaN: poisoned prefix block -> keep)
            } else {
                const double mc = S1 / m;  // mean - c_cur
                const double var = S2 / m - mc * mc, sg = sqrt(var), mean = c_cur + mc;
                kvm_stat_guard gd;
                const double hist = kvm_run_hist_of(qd, cur.run);
                gd.mean_err = E1 / m + kvm_hist_mean(hist, m);
                const double var_err = E2 / m + 2.0 * fabs(mc) * gd.mean_err + kvm_hist_var(hist, m);
                gd.g = kvm_sigma_guard(mean, sg, var_err / (2.0 * var) + gd.mean_err / sg);
                gd.std_hi = sqrt(fmax(var, 0.0) + var_err + 1e-8 * (mean * mean + fabs(var)));
                passed = kvm_cnsm_filter_guarded(mean, sg, qd.meanQ, qd.stdQ, qd.alpha, qd.beta, gd);
                if (passed) {
                    if (!(sg == sg) || !(sg > 0.0) || kvm_guard_degenerate(gd.g)) {
                        keep = true;  // degenerate sigma: the exact pass decides
                    } else {
                        const double inv = 1.0 / sg;
                        const double xx = S2 - (2.0 * mc) * S1 + m * (mc * mc);
                        const double xz = r1 + qbar * S1 - mc * (qtot.x + m * qbar);
                        const double zz = qtot.y + (2.0 * qbar) * qtot.x + m * (qbar * qbar);
                        const double slack =
                            eps2 * 1e-9 + kvm_eps2_slack(eps2, m, gd.g, 4.0) +
                            1e-11 * (fabs(xx) * inv * inv + 2.0 * (fabs(r1) + fabs(qbar * S1) + fabs(mc * qtot.x)) * inv + fabs(zz)) +
                            (E2 + 2.0 * fabs(mc) * E1) * inv * inv + 2.0 * inv * fabs(qbar) * E1;
                        keep = !((xx * inv * inv - 2.0 * xz * inv + zz) > eps2 + slack);
                    }
                }
            }
        }
        if (NORM) {
            const unsigned long long pm = __ballot(passed);
            if (lane == 0 && pm) atomicAdd(&A.counters[KVM_CNT_PASS_FILTER], (unsigned long long)__popcll(pm));
        }
        if (keep) {
            const unsigned long long idx = atomicAdd(&A.counters[KVM_CNT_CANDS1], 1ULL);
            if ((int64_t)idx < A.cap) A.surv[idx] = kvm_surv{cur.start + mycand, cur.run};
        }
        cur.start = __builtin_amdgcn_readfirstlane(nxt_v.start);
        cur.count = __builtin_amdgcn_readfirstlane(nxt_v.count);
        cur.run = __builtin_amdgcn_readfirstlane(nxt_v.run);
        c_cur = c_nxt;
    }
}

hipError_t kvm_launch_ed_jl(const double *series, const double2 *prefix, int64_t n, const kvm_query_dev &qd,
                            const kvm_tile *tiles, int64_t n_tiles, kvm_surv *surv, int64_t cap,
                            unsigned long long *counters, int num_cu, hipStream_t s) {
    (void)n;
    if (n_tiles <= 0) return hipSuccess;
    int64_t blocks = (int64_t)num_cu * 2;  // 2 blocks x 4 waves per CU = 2 waves per SIMD (register-limited)
    if (blocks * 4 > n_tiles) blocks = (n_tiles + 3) / 4;
    jl_args A{series, prefix, qd, surv, cap, counters, tiles, n_tiles};
    if (qd.normalised)
        hipLaunchKernelGGL((kvm_ed_jl_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, A);
    else
        hipLaunchKernelGGL((kvm_ed_jl_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s, A);
    return hipGetLastError();
}
