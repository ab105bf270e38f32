// kvm_ed_dot.hip -- correlation-form ED pruner (RSM-ED, cNSM-ED) for gfx950: ONE v_fma_f64 per (candidate, point).
//
// The exact pass (kvm_exact.hip) decides membership and computes the reference's distance for every survivor, so the
// pruning kernel may evaluate the distance in any algebraically equivalent form as long as it never drops a
// candidate whose true value is within the rounding uncertainty of the threshold.  This kernel uses
//     RSM : sum (x-q)^2            = S2' - 2 (acc + d S1') + T2 + 2 d T1 + m d^2
//     cNSM: sum ((x-mu)/sg - zq)^2 = m   - (2/sg) (acc - S1' T1 / m) + Szq2
// with  x' = x - c (c = first point of the tile: removes the level of the data), qc = q - mean(q),
//       acc = sum x'_j qc_j  (the only O(m) work: one FMA per term),
//       S1' = sum x', S2' = sum x'^2 (O(1) from the shifted block prefix sums), T1 = sum qc, T2 = sum qc^2, d = mean(q) - c.
// Everything is formed around local levels, so the terms are O(m var) and the cancellation the naive dot-product
// identity suffers from is bounded by the explicit guard band 1e-11 * (sum of the magnitudes of the
// terms) -- about 1e5 ulps, versus the <= sqrt(m) ulps the accumulations can lose.  Candidates inside the band go to
// the exact pass like every other survivor.
//
// Mapping: one wavefront = 8 rows x 8 lanes; a row is a tile of <= 32 consecutive candidates, a lane owns 4 adjacent
// candidates.  Per 2 points a lane issues one ds_read_b128 of the window (2 new values) and one broadcast
// ds_read_b128 of (qc[j], qc[j+1]) for 8 FMAs.  Lane stride in a row is 32 B, so rows {2,3,6,7} are displaced by 16 B
// against rows {0,1,4,5}: every 16-lane group of ds_read_b128 then covers 16 distinct 16-B slots (conflict-free).
// All LDS regions are wave-private: no workgroup barrier.
#include "kvm_internal.h"

#define ED4_ROWS 8
#define ED4_ROWLEN (ED4_JC + 32)              // staged points per row per chunk
#define ED4_LOADS (ED4_ROWLEN / 64)
#define ED4_RS (ED4_ROWLEN + 32)              // row stride in doubles (x8 B = multiple of 256 B), room for the 16-B shift
#define ED4_WAVE_LDS (ED4_ROWS * ED4_RS + ED4_ROWLEN)
static_assert(ED4_ROWLEN % 64 == 0 && (ED4_RS * 8) % 256 == 0 && ED4_JC % 8 == 0 && (ED4_JC / 4) % 2 == 0, "layout");
static_assert(4 * ED4_WAVE_LDS * 8 <= 65536 && ED4_WAVE_LDS % 2 == 0 && ED4_RS % 2 == 0, "static LDS");

// 16-B LDS loads are written with a clang ext_vector type: loads of HIP's double2 struct get scalarised and re-paired
// into ds_read2_b64 (8-B banking rules => 4-way conflicts on this layout); <2 x double> loads stay ds_read_b128.
typedef double ed4_d2 __attribute__((ext_vector_type(2)));

template <bool NORM, bool ABANDON>
__global__ __launch_bounds__(256, 2) void kvm_ed_dot_kernel(const double *__restrict__ series, int64_t n,
                                                            const double2 *__restrict__ prefix, kvm_query_dev qd,
                                                            const kvm_tile *__restrict__ tiles, int64_t n_tiles,
                                                            kvm_surv *__restrict__ surv, int64_t cap,
                                                            unsigned long long *__restrict__ counters) {
    // declared as double2 so that the window/query reads are provably 16-B aligned (ds_read_b128, not ds_read2_b64)
    __shared__ __attribute__((aligned(256))) double2 lds_all2[4 * ED4_WAVE_LDS / 2];
    double *lds_all = reinterpret_cast<double *>(lds_all2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane >> 3, lc = lane & 7;
    const int m = qd.m;
    const double eps2 = qd.eps2;
    const double *__restrict__ qc = qd.qc;
    double *wlds = lds_all + wave * ED4_WAVE_LDS;
    double *qlds = wlds + ED4_ROWS * ED4_RS;
    const int nchunks = (m + ED4_JC - 1) / ED4_JC;

    const int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * ED4_ROWS;
    if (t0 >= n_tiles) return;
    // wave-uniform row descriptors (for staging) ...
    int32_t rstart[ED4_ROWS];
    double rshift[ED4_ROWS];
#pragma unroll
    for (int r = 0; r < ED4_ROWS; r++) {
        rstart[r] = (t0 + r < n_tiles) ? tiles[t0 + r].start : 1;  // empty rows stage from the head of the series
        rshift[r] = series[rstart[r] - 1];
    }
    // ... and this lane's own row
    kvm_tile mine;
    mine.start = 1; mine.count = 0; mine.run_start = 1; mine.pad = 0;
    if (t0 + row < n_tiles) mine = tiles[t0 + row];
    const double c_row = series[mine.start - 1];
    const int c0 = 4 * lc;
    const double T1 = qd.q_pref[nchunks].x, T2 = qd.q_pref[nchunks].y, qbar = qd.q_mean;

    bool v[4];
    double S1[4], S2[4];       // shifted window sums re-based to c_row
    double mu_c[4], sg[4];     // cNSM: mu - c_row, sigma
    double lim[4];             // pruning threshold on the correlation-form distance (eps^2 + guard band)
    double gstd[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        v[a] = c0 + a < mine.count;
        S1[a] = S2[a] = 0.0;
        mu_c[a] = 0.0; sg[a] = 1.0; lim[a] = 0.0; gstd[a] = 0.0;
        if (v[a]) {
            double s1, s2, cb;
            kvm_window_shifted_sums(prefix, series, (int64_t)mine.start - 1 + c0 + a, m, s1, s2, cb);
            const double dc = cb - c_row;
            S1[a] = s1 + m * dc;
            S2[a] = s2 + (2.0 * dc) * s1 + m * (dc * dc);
            if (NORM) {
                const double ms = S1[a] / m;
                mu_c[a] = ms;
                sg[a] = sqrt(S2[a] / m - ms * ms);
                const double mean = c_row + ms;
                gstd[a] = kvm_sigma_guard(mean, sg[a]);
                v[a] = kvm_cnsm_filter_guarded(mean, sg[a], qd.meanQ, qd.stdQ, qd.alpha, qd.beta, gstd[a]);
                if (!(sg[a] == sg[a]) || !(sg[a] > 0.0)) { sg[a] = 0.0; lim[a] = __builtin_huge_val(); }  // exact pass decides
            }
        }
    }
    if (NORM) {
        unsigned cntp = 0;
#pragma unroll
        for (int a = 0; a < 4; a++) cntp += (unsigned)__popcll(__ballot(v[a]));
        if (lane == 0 && cntp) atomicAdd(&counters[KVM_CNT_PASS_FILTER], (unsigned long long)cntp);
        if (cntp == 0) return;  // nothing in this wave passed the (alpha, beta) filter
    }

    double stage[ED4_ROWS + 1][ED4_LOADS];
    auto load_chunk = [&](int j0) {
#pragma unroll
        for (int r = 0; r < ED4_ROWS; r++) {
            const double *src = series + ((int64_t)rstart[r] - 1 + j0);  // unguarded: series is padded (KVM_SERIES_PAD)
#pragma unroll
            for (int t = 0; t < ED4_LOADS; t++) stage[r][t] = src[t * 64 + lane] - rshift[r];
        }
#pragma unroll
        for (int t = 0; t < ED4_LOADS; t++) stage[ED4_ROWS][t] = qc[j0 + t * 64 + lane];
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int r = 0; r < ED4_ROWS; r++) {
#pragma unroll
            for (int t = 0; t < ED4_LOADS; t++) wlds[r * ED4_RS + ((r >> 1) & 1) * 2 + t * 64 + lane] = stage[r][t];
        }
#pragma unroll
        for (int t = 0; t < ED4_LOADS; t++) qlds[t * 64 + lane] = stage[ED4_ROWS][t];
    };

    // correlation-form distance after J points (J == m: the whole window) and its guard band
    auto dist_form = [&](int a, double acc, int J, const double2 qp, double s1J, double s2J, double &slack) -> double {
        if (!NORM) {
            const double d = qbar - c_row;
            const double cross = acc + d * s1J;
            const double qq = qp.y + (2.0 * d) * qp.x + J * (d * d);
            slack = eps2 * 1e-9 + 1e-11 * (fabs(s2J) + 2.0 * fabs(acc) + 2.0 * fabs(d * s1J) + fabs(qq));
            return s2J - 2.0 * cross + qq;
        } else {
            // sum_J ((x-mu)/sg - zq)^2, zq = qc + qbar
            const double mc = mu_c[a], s = sg[a];
            const double xx = s2J - (2.0 * mc) * s1J + J * (mc * mc);                 // sum (x-mu)^2
            const double xz = acc + qbar * s1J - mc * (qp.x + J * qbar);             // sum (x-mu) zq
            const double zz = qp.y + (2.0 * qbar) * qp.x + J * (qbar * qbar);        // sum zq^2
            const double inv = 1.0 / s;
            slack = eps2 * 1e-9 + kvm_eps2_slack(eps2, m, gstd[a], 4.0) +
                    1e-11 * (fabs(xx) * inv * inv + 2.0 * (fabs(acc) + fabs(qbar * s1J) + fabs(mc * qp.x)) * inv + fabs(zz));
            return xx * inv * inv - 2.0 * xz * inv + zz;
        }
    };

    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    load_chunk(0);
    store_chunk();
    const ed4_d2 *X2 = reinterpret_cast<const ed4_d2 *>(lds_all2) + (wave * (ED4_WAVE_LDS / 2) + row * (ED4_RS / 2) + ((row >> 1) & 1) + 2 * lc);
    const ed4_d2 *Q2 = reinterpret_cast<const ed4_d2 *>(lds_all2) + (wave * (ED4_WAVE_LDS / 2) + ED4_ROWS * (ED4_RS / 2));
    for (int ch = 0; ch < nchunks; ch++) {
        const int j0 = ch * ED4_JC;
        const bool more = ch + 1 < nchunks;
        if (more) load_chunk(j0 + ED4_JC);
        const int jlim = min(ED4_JC, m - j0);
        ed4_d2 wa = X2[0], wb = X2[1];  // X[c0+0..3]
        if (jlim == ED4_JC) {
            // 4 points per group; the LDS reads of group g+1 are issued before the 16 FMAs of group g (all DS
            // operations return in order, so the compiler can wait with a counted lgkmcnt).
#define ED4_FMAS(n0, n1, q0, q1)                                                                       \
    do {                                                                                              \
        acc[0] = __builtin_fma(wa.x, q0.x, acc[0]); acc[1] = __builtin_fma(wa.y, q0.x, acc[1]);       \
        acc[2] = __builtin_fma(wb.x, q0.x, acc[2]); acc[3] = __builtin_fma(wb.y, q0.x, acc[3]);       \
        acc[0] = __builtin_fma(wa.y, q0.y, acc[0]); acc[1] = __builtin_fma(wb.x, q0.y, acc[1]);       \
        acc[2] = __builtin_fma(wb.y, q0.y, acc[2]); acc[3] = __builtin_fma(n0.x, q0.y, acc[3]);       \
        acc[0] = __builtin_fma(wb.x, q1.x, acc[0]); acc[1] = __builtin_fma(wb.y, q1.x, acc[1]);       \
        acc[2] = __builtin_fma(n0.x, q1.x, acc[2]); acc[3] = __builtin_fma(n0.y, q1.x, acc[3]);       \
        acc[0] = __builtin_fma(wb.y, q1.y, acc[0]); acc[1] = __builtin_fma(n0.x, q1.y, acc[1]);       \
        acc[2] = __builtin_fma(n0.y, q1.y, acc[2]); acc[3] = __builtin_fma(n1.x, q1.y, acc[3]);       \
        wa = n0;                                                                                      \
        wb = n1;                                                                                      \
    } while (0)
            ed4_d2 a0 = X2[2], a1 = X2[3], aq0 = Q2[0], aq1 = Q2[1];
#pragma unroll 1
            for (int g = 0; g + 2 < ED4_JC / 4; g += 2) {
                const ed4_d2 b0 = X2[2 * g + 4], b1 = X2[2 * g + 5], bq0 = Q2[2 * g + 2], bq1 = Q2[2 * g + 3];
                ED4_FMAS(a0, a1, aq0, aq1);
                a0 = X2[2 * g + 6]; a1 = X2[2 * g + 7]; aq0 = Q2[2 * g + 4]; aq1 = Q2[2 * g + 5];
                ED4_FMAS(b0, b1, bq0, bq1);
            }
            {   // last two groups (ED4_JC / 4 is even)
                constexpr int g = ED4_JC / 4 - 2;
                const ed4_d2 b0 = X2[2 * g + 4], b1 = X2[2 * g + 5], bq0 = Q2[2 * g + 2], bq1 = Q2[2 * g + 3];
                ED4_FMAS(a0, a1, aq0, aq1);
                ED4_FMAS(b0, b1, bq0, bq1);
            }
#undef ED4_FMAS
        } else {
            const double *X = reinterpret_cast<const double *>(X2);
            for (int jj = 0; jj < jlim; jj++) {
                const double qv = qlds[jj];
#pragma unroll
                for (int a = 0; a < 4; a++) acc[a] = __builtin_fma(X[jj + a], qv, acc[a]);
            }
        }
        if (more) {
            // early abandon (reference behaviour): partial sums of squares only grow, so a candidate whose distance
            // over the first J points already exceeds the pruning threshold is out.  Checked after the first chunk
            // (random candidates die here) and every 8 chunks afterwards.
            if (ABANDON && (ch == 0 || (ch & 7) == 7)) {
                const int J = j0 + ED4_JC;
                const double2 qp = qd.q_pref[ch + 1];
                bool alive = false;
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    if (v[a]) {
                        double s1, s2, cb, slack;
                        kvm_window_shifted_sums(prefix, series, (int64_t)mine.start - 1 + c0 + a, J, s1, s2, cb);
                        const double dc = cb - c_row;
                        const double s1J = s1 + J * dc, s2J = s2 + (2.0 * dc) * s1 + J * (dc * dc);
                        const double dj = dist_form(a, acc[a], J, qp, s1J, s2J, slack);
                        if (dj > eps2 + slack && lim[a] == 0.0) v[a] = false;
                        alive |= v[a];
                    }
                }
                if (!__any(alive)) return;
            }
            store_chunk();
        }
    }
    const double2 qtot = make_double2(T1, T2);
#pragma unroll
    for (int a = 0; a < 4; a++) {
        if (v[a]) {
            double slack;
            const double d2 = dist_form(a, acc[a], m, qtot, S1[a], S2[a], slack);
            if (lim[a] != 0.0 || d2 <= eps2 + slack) {
                unsigned long long idx = atomicAdd(&counters[KVM_CNT_CANDS1], 1ULL);
                if ((int64_t)idx < cap) surv[idx] = kvm_surv{mine.start + c0 + a, mine.run_start};
            }
        }
    }
}

hipError_t kvm_launch_ed_dot(const double *series, const double2 *prefix, int64_t n, const kvm_query_dev &qd,
                             const kvm_tile *tiles, int64_t n_tiles, kvm_surv *surv, int64_t cap,
                             unsigned long long *counters, hipStream_t s) {
    if (n_tiles <= 0) return hipSuccess;
    const int64_t blocks = (n_tiles + 4 * ED4_ROWS - 1) / (4 * ED4_ROWS);
    dim3 grid((unsigned)blocks), block(256);
#define LAUNCH(NORM, AB)                                                                                       \
    hipLaunchKernelGGL((kvm_ed_dot_kernel<NORM, AB>), grid, block, 0, s, series, n, prefix, qd, tiles, n_tiles, \
                       surv, cap, counters)
    if (qd.normalised) {
        if (qd.early_abandon) LAUNCH(true, true); else LAUNCH(true, false);
    } else {
        if (qd.early_abandon) LAUNCH(false, true); else LAUNCH(false, false);
    }
#undef LAUNCH
    return hipGetLastError();
}
