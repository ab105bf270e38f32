// kvm_dtw_lb.hip -- LDS-tiled first stage of the DTW cascade for gfx950: (alpha, beta) filter, LB_Kim and LB_Keogh on the
// query envelope for EVERY candidate window, fused in one kernel.
//
// Replaces, for the bulk of the candidates, K/NormQueryEngineDtw.java:550-563 / K/QueryEngineDtw.java:409-419:
//   mean/std (prefix sums) -> |mean-meanQ| <= beta, std/stdQ in [1/alpha, alpha] -> lbKimHierarchy (K/utils/DtwUtils.java:
//   149-189) -> lbKeoghCumulative (:206-222).
// LB_Keogh is an O(m) sum over the window, like ED, and consecutive candidates overlap in all but one point, so it is
// tiled exactly like the correlation-form ED kernel (kvm_ed_dot.hip): one wavefront = 8 rows x 8 lanes, a row = one tile
// of <= 32 consecutive candidates staged once in wave-private LDS, a lane = 4 adjacent candidates; the query envelope
// (upper/lower, wave-uniform) is staged next to the rows and read with broadcast ds_read_b128.  Per (candidate, point):
//     v = x * inv_sigma - mean * inv_sigma;  t = max(v - U_j, L_j - v, 0);  lb += t * t         (6 VALU ops; RSM: 5)
// against 3 LDS reads per 8 terms, i.e. VALU-bound.  A wave stops as soon as every candidate it holds has exceeded the
// pruning threshold (lower bounds only grow).  All values here use the approximate mean/std of the prefix sums: the
// threshold carries the guard band of kvm_internal.h and survivors are re-examined downstream (LB_Keogh on the data
// envelope, replayed mean/std, exact DP), so this kernel can only prune candidates that no arithmetic would accept.
#include "kvm_internal.h"

#define LB4_ROWS 8
#define LB4_JC 96
#define LB4_ROWLEN (LB4_JC + 32)
#define LB4_LOADS (LB4_ROWLEN / 64)
#define LB4_RS (LB4_ROWLEN + 32)
#define LB4_WAVE_LDS (LB4_ROWS * LB4_RS + 2 * LB4_ROWLEN)  // 8 rows + upper + lower envelope chunk
static_assert(LB4_ROWLEN % 64 == 0 && (LB4_RS * 8) % 256 == 0 && (LB4_JC / 4) % 2 == 0, "layout");
static_assert(4 * LB4_WAVE_LDS * 8 <= 65536 && LB4_WAVE_LDS % 2 == 0, "static LDS");

typedef double lb4_d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double lb4_dsq(double a, double b) { return (a - b) * (a - b); }

// LB_Kim, K/utils/DtwUtils.java:149-189 (the full bound, no early returns); tt = first point of the window.
__device__ __forceinline__ double lb4_kim(const double *__restrict__ tt, const double *__restrict__ q, int m, double mu, double s) {
    double d, l;
    const double x0 = (tt[0] - mu) / s, y0 = (tt[m - 1] - mu) / s;
    l = lb4_dsq(x0, q[0]) + lb4_dsq(y0, q[m - 1]);
    const double x1 = (tt[1] - mu) / s;
    d = fmin(lb4_dsq(x1, q[0]), lb4_dsq(x0, q[1]));
    d = fmin(d, lb4_dsq(x1, q[1]));
    l += d;
    const double y1 = (tt[m - 2] - mu) / s;
    d = fmin(lb4_dsq(y1, q[m - 1]), lb4_dsq(y0, q[m - 2]));
    d = fmin(d, lb4_dsq(y1, q[m - 2]));
    l += d;
    const double x2 = (tt[2] - mu) / s;
    d = fmin(lb4_dsq(x0, q[2]), lb4_dsq(x1, q[2]));
    d = fmin(d, lb4_dsq(x2, q[2]));
    d = fmin(d, lb4_dsq(x2, q[1]));
    d = fmin(d, lb4_dsq(x2, q[0]));
    l += d;
    const double y2 = (tt[m - 3] - mu) / s;
    d = fmin(lb4_dsq(y0, q[m - 3]), lb4_dsq(y1, q[m - 3]));
    d = fmin(d, lb4_dsq(y2, q[m - 3]));
    d = fmin(d, lb4_dsq(y2, q[m - 2]));
    d = fmin(d, lb4_dsq(y2, q[m - 1]));
    l += d;
    return l;
}

// (alpha, beta) filter + LB_Kim of one candidate with the pruning arithmetic (prefix-sum mean/std, guard bands); also
// returns the values stage 1 keeps per candidate.  `w0` = 0-based window start.
template <bool NORM>
__device__ __forceinline__ void lb4_prefilter(const double *__restrict__ series, const double2 *__restrict__ prefix,
                                              const kvm_query_dev &qd, int64_t w0, bool &pass_filter, bool &pass_kim,
                                              double &mean, double &sd, double &inv, double &nmi, double &lim) {
    const int m = qd.m;
    mean = 0.0; sd = 1.0; inv = 1.0; nmi = 0.0;
    lim = qd.eps2 * (1.0 + 1e-9);
    pass_filter = true;
    if (NORM) {
        const kvm_stat_guard sg = kvm_window_mean_std_g(prefix, series, w0, m, mean, sd);
        pass_filter = kvm_cnsm_filter_guarded(mean, sd, qd.meanQ, qd.stdQ, qd.alpha, qd.beta, sg);
        lim = qd.eps2 + kvm_eps2_slack(qd.eps2, m, sg.g, 16.0);
        if (!(sd == sd) || !(sd > 0.0) || kvm_guard_degenerate(sg.g)) {  // degenerate sigma: no bound, the exact pass decides
            sd = __builtin_nan("");
            lim = __builtin_huge_val();
            inv = 0.0;
            nmi = 0.0;
        } else {
            inv = 1.0 / sd;
            nmi = -mean * inv;
        }
    }
    pass_kim = pass_filter;
    if (pass_filter && m >= 6 && sd == sd) pass_kim = lb4_kim(series + w0, qd.q, m, mean, sd) <= lim;
}

template <bool NORM>
__global__ __launch_bounds__(256, 2) void kvm_dtw_lb_tiled_kernel(const double *__restrict__ series,
                                                                  const double2 *__restrict__ prefix, kvm_query_dev qd,
                                                                  const kvm_tile *__restrict__ tiles, int64_t n_tiles,
                                                                  kvm_cand *__restrict__ out,
                                                                  unsigned long long *__restrict__ counters,
                                                                  const unsigned long long *__restrict__ d_n_tiles,
                                                                  int64_t split_units) {
    __shared__ __attribute__((aligned(256))) double2 lds_all2[4 * LB4_WAVE_LDS / 2];
    if (d_n_tiles) n_tiles = (int64_t)*d_n_tiles;  // tile list built on the device: the launch only knows a bound
    double *lds_all = reinterpret_cast<double *>(lds_all2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane >> 3, lc = lane & 7;
    const int m = qd.m;
    double *wlds = lds_all + wave * LB4_WAVE_LDS;
    double *ulds = wlds + LB4_ROWS * LB4_RS, *llds = ulds + LB4_ROWLEN;
    const int nchunks = (m + LB4_JC - 1) / LB4_JC;

    // Window split (as in kvm_ed_dot.hip): when the tile list is short -- a query with few answers leaves a few hundred
    // candidates, and a lone wavefront needs ~0.25 ms for the m points of one that survives to the end -- the four waves
    // of a workgroup share one unit of 8 tiles, each summing a quarter of the chunks (lower bounds are sums of
    // non-negative terms); parts 1..3 hand their partial sums to part 0 through LDS behind one workgroup barrier.
    const int64_t units = (n_tiles + LB4_ROWS - 1) / LB4_ROWS;
    const int f = units <= split_units ? 4 : 1;
    const int64_t slot = (int64_t)blockIdx.x * 4 + wave;
    const int part = (int)(slot % f);
    const int64_t t0 = (slot / f) * LB4_ROWS;
    if (t0 >= n_tiles) return;  // (f == 4: the whole workgroup leaves; f == 1: no barrier below)
    int32_t rstart[LB4_ROWS];
#pragma unroll
    for (int r = 0; r < LB4_ROWS; r++) rstart[r] = (t0 + r < n_tiles) ? tiles[t0 + r].start : 1;
    kvm_tile mine;
    mine.start = 1; mine.count = 0; mine.run_start = 1; mine.pad = 0;
    if (t0 + row < n_tiles) mine = tiles[t0 + row];
    const int c0 = 4 * lc;

    // ---- per candidate: mean/std, (alpha, beta) filter, LB_Kim ----
    bool v[4];
    double mean[4], sd[4], inv[4], nmi[4], lim[4], lb[4];
    bool work = true;  // wave-uniform
    unsigned n_filter = 0, n_kim = 0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        v[a] = c0 + a < mine.count;
        mean[a] = 0.0; sd[a] = 1.0; inv[a] = 1.0; nmi[a] = 0.0; lb[a] = 0.0;
        lim[a] = qd.eps2 * (1.0 + 1e-9);
        if (v[a]) {
            bool pf, pk;
            lb4_prefilter<NORM>(series, prefix, qd, (int64_t)mine.start - 1 + c0 + a, pf, pk, mean[a], sd[a], inv[a], nmi[a],
                                lim[a]);
            n_filter += pf;
            n_kim += pk;
            v[a] = pk;
        }
    }
    {
        // wave totals of the two counters (one atomic each)
        unsigned f = n_filter, k = n_kim;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            f += __shfl_xor(f, o, 64);
            k += __shfl_xor(k, o, 64);
        }
        if (lane == 0 && !d_n_tiles && part == 0) {  // (with device-built tiles the retile kernel has counted every candidate)
            if (NORM && f) atomicAdd(&counters[KVM_CNT_PASS_FILTER], (unsigned long long)f);
            if (k) atomicAdd(&counters[KVM_CNT_PASS_KIM], (unsigned long long)k);
        }
        if (k == 0 && f == 1) return;  // nothing left in this wave
        work = k != 0;
    }

    // ---- LB_Keogh on the query envelope, tiled ----
    double stage[LB4_ROWS + 2][LB4_LOADS];
    auto load_chunk = [&](int j0) {
#pragma unroll
        for (int r = 0; r < LB4_ROWS; r++) {
            const double *src = series + ((int64_t)rstart[r] - 1 + j0);  // unguarded: series is padded (KVM_SERIES_PAD)
#pragma unroll
            for (int t = 0; t < LB4_LOADS; t++) stage[r][t] = src[t * 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < LB4_LOADS; t++) {
            stage[LB4_ROWS][t] = qd.qu[j0 + t * 64 + lane];      // (query block is padded as well)
            stage[LB4_ROWS + 1][t] = qd.ql[j0 + t * 64 + lane];
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int r = 0; r < LB4_ROWS; r++) {
#pragma unroll
            for (int t = 0; t < LB4_LOADS; t++) wlds[r * LB4_RS + ((r >> 1) & 1) * 2 + t * 64 + lane] = stage[r][t];
        }
#pragma unroll
        for (int t = 0; t < LB4_LOADS; t++) {
            ulds[t * 64 + lane] = stage[LB4_ROWS][t];
            llds[t * 64 + lane] = stage[LB4_ROWS + 1][t];
        }
    };
#define LB4_TERM(a, xv, hi, lo)                                                       \
    do {                                                                              \
        const double z_ = NORM ? __builtin_fma((xv), inv[a], nmi[a]) : (xv);          \
        double t_ = fmax(z_ - (hi), (lo) - z_);                                       \
        t_ = fmax(t_, 0.0);                                                           \
        lb[a] = __builtin_fma(t_, t_, lb[a]);                                         \
    } while (0)

    const int ch_lo = nchunks * part / f, ch_hi = nchunks * (part + 1) / f;
    if (work && ch_lo < ch_hi) {
        load_chunk(ch_lo * LB4_JC);
        store_chunk();
    }
    const lb4_d2 *X2 = reinterpret_cast<const lb4_d2 *>(lds_all2) +
                       (wave * (LB4_WAVE_LDS / 2) + row * (LB4_RS / 2) + ((row >> 1) & 1) + 2 * lc);
    const lb4_d2 *U2 = reinterpret_cast<const lb4_d2 *>(lds_all2) + (wave * (LB4_WAVE_LDS / 2) + LB4_ROWS * (LB4_RS / 2));
    const lb4_d2 *L2 = U2 + LB4_ROWLEN / 2;
    for (int ch = ch_lo; work && ch < ch_hi; ch++) {
        const int j0 = ch * LB4_JC;
        const bool more = ch + 1 < ch_hi;
        if (more) load_chunk(j0 + LB4_JC);
        const int jlim = min(LB4_JC, m - j0);
        if (jlim == LB4_JC) {
            lb4_d2 wa = X2[0], wb = X2[1];
#pragma unroll 2
            for (int g = 0; g < LB4_JC / 4; g++) {
                const lb4_d2 n0 = X2[2 * g + 2], n1 = X2[2 * g + 3];
                const lb4_d2 u0 = U2[2 * g], u1 = U2[2 * g + 1], l0 = L2[2 * g], l1 = L2[2 * g + 1];
                LB4_TERM(0, wa.x, u0.x, l0.x); LB4_TERM(1, wa.y, u0.x, l0.x); LB4_TERM(2, wb.x, u0.x, l0.x); LB4_TERM(3, wb.y, u0.x, l0.x);
                LB4_TERM(0, wa.y, u0.y, l0.y); LB4_TERM(1, wb.x, u0.y, l0.y); LB4_TERM(2, wb.y, u0.y, l0.y); LB4_TERM(3, n0.x, u0.y, l0.y);
                LB4_TERM(0, wb.x, u1.x, l1.x); LB4_TERM(1, wb.y, u1.x, l1.x); LB4_TERM(2, n0.x, u1.x, l1.x); LB4_TERM(3, n0.y, u1.x, l1.x);
                LB4_TERM(0, wb.y, u1.y, l1.y); LB4_TERM(1, n0.x, u1.y, l1.y); LB4_TERM(2, n0.y, u1.y, l1.y); LB4_TERM(3, n1.x, u1.y, l1.y);
                wa = n0;
                wb = n1;
            }
        } else {
            const double *X = reinterpret_cast<const double *>(X2);
            for (int jj = 0; jj < jlim; jj++) {
                const double hi = ulds[jj], lo = llds[jj];
                LB4_TERM(0, X[jj], hi, lo); LB4_TERM(1, X[jj + 1], hi, lo); LB4_TERM(2, X[jj + 2], hi, lo); LB4_TERM(3, X[jj + 3], hi, lo);
            }
        }
        if (more) {
            bool alive = false;
#pragma unroll
            for (int a = 0; a < 4; a++) alive |= v[a] && lb[a] <= lim[a];
            if (!__any(alive)) {  // lower bounds only grow: this part's own sum already rules every candidate out
                if (f == 1) return;
                break;
            }
            store_chunk();
        }
    }
#undef LB4_TERM
    if (f > 1) {
        if (part > 0 && work) {
#pragma unroll
            for (int a = 0; a < 4; a++) wlds[a * 64 + lane] = lb[a];
        }
        __syncthreads();
        if (part > 0 || !work) return;
        for (int p = 1; p < f; p++) {
#pragma unroll
            for (int a = 0; a < 4; a++) lb[a] += wlds[p * LB4_WAVE_LDS + a * 64 + lane];
        }
    }
    // survivors of the filter, LB_Kim and LB_Keogh(query envelope): wave-aggregated append
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const bool keep = v[a] && lb[a] <= lim[a];
        const unsigned long long bal = __ballot(keep);
        if (bal == 0ULL) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&counters[KVM_CNT_CANDS1], (unsigned long long)__popcll(bal));
        base = __shfl(base, 0, 64);
        if (keep) {
            kvm_cand c;
            c.start = mine.start + c0 + a;
            c.run_start = mine.run_start;
            c.mean = mean[a];
            c.std = sd[a];
            c.lim = lim[a];
            out[base + __popcll(bal & ((1ULL << lane) - 1ULL))] = c;
        }
    }
}

// DTW retile: the (alpha, beta) filter and LB_Kim for EVERY candidate, one lane each (64 candidates of a tile in
// parallel: stage 1 above evaluates them four per lane, one after the other, which made it the slowest kernel of a
// query with few answers), then every 64-candidate tile is trimmed to the span of its survivors, cut into <= 32-candidate
// tiles for stage 1 and appended to a device-side list (64 tiles per workgroup collected in LDS, one global atomic).
template <bool NORM>
__global__ __launch_bounds__(256) void kvm_retile_dtw_kernel(const double *__restrict__ series,
                                                             const double2 *__restrict__ prefix, kvm_query_dev qd,
                                                             const kvm_tile *__restrict__ tiles, int64_t n_tiles,
                                                             kvm_tile *__restrict__ out_tiles,
                                                             unsigned long long *__restrict__ counters) {
    __shared__ kvm_tile outs[2 * 64];
    __shared__ int n_out;
    __shared__ unsigned n_f, n_k;
    __shared__ unsigned long long base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        n_out = 0;
        n_f = 0;
        n_k = 0;
    }
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * 64 + wave * 16;
    unsigned cf = 0, ck = 0;
    for (int i = 0; i < 16; i++) {
        const int64_t t = t0 + i;
        if (t >= n_tiles) break;  // wave-uniform
        const kvm_tile tl = tiles[t];
        bool pf = false, pk = false;
        if (lane < tl.count) {
            double mean, sd, inv, nmi, lim;
            lb4_prefilter<NORM>(series, prefix, qd, (int64_t)tl.start - 1 + lane, pf, pk, mean, sd, inv, nmi, lim);
        }
        const unsigned long long mf = __ballot(pf), mk = __ballot(pk);
        cf += (unsigned)__popcll(mf);
        ck += (unsigned)__popcll(mk);
        if (mk != 0 && lane == 0) {
            const int first = __ffsll((long long)mk) - 1, last = 63 - __clzll((long long)mk);
            kvm_tile o;
            o.run_start = tl.run_start;
            o.pad = 0;
            o.start = tl.start + first;
            o.count = min(32, last - first + 1);
            outs[atomicAdd(&n_out, 1)] = o;
            if (last - first + 1 > 32) {
                o.start = tl.start + first + 32;
                o.count = last - first + 1 - 32;
                outs[atomicAdd(&n_out, 1)] = o;
            }
        }
    }
    if (lane == 0) {
        if (cf) atomicAdd(&n_f, cf);
        if (ck) atomicAdd(&n_k, ck);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        base = n_out ? atomicAdd(&counters[KVM_CNT_TILES_A], (unsigned long long)n_out) : 0;
        if (NORM && n_f) atomicAdd(&counters[KVM_CNT_PASS_FILTER], (unsigned long long)n_f);
        if (n_k) atomicAdd(&counters[KVM_CNT_PASS_KIM], (unsigned long long)n_k);
    }
    __syncthreads();
    if ((int)threadIdx.x < n_out) out_tiles[base + threadIdx.x] = outs[threadIdx.x];
}

hipError_t kvm_launch_retile_dtw(const double *series, const double2 *prefix, const kvm_query_dev &qd, const kvm_tile *tiles,
                                 int64_t n_tiles, kvm_tile *out_tiles, unsigned long long *counters, hipStream_t s) {
    if (n_tiles <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_tiles + 63) / 64));
    if (qd.normalised)
        hipLaunchKernelGGL((kvm_retile_dtw_kernel<true>), grid, dim3(256), 0, s, series, prefix, qd, tiles, n_tiles, out_tiles,
                           counters);
    else
        hipLaunchKernelGGL((kvm_retile_dtw_kernel<false>), grid, dim3(256), 0, s, series, prefix, qd, tiles, n_tiles, out_tiles,
                           counters);
    return hipGetLastError();
}

// `d_n_tiles` != nullptr: `tiles` was built on the device and holds *d_n_tiles entries; `n_tiles` is then an upper bound
// and the kernel splits the window over the four waves of a workgroup when the real list is short.
hipError_t kvm_launch_dtw_lb_tiled(const double *series, const double2 *prefix, int64_t n, const kvm_query_dev &qd,
                                   const kvm_tile *tiles, int64_t n_tiles, kvm_cand *out, unsigned long long *counters,
                                   const unsigned long long *d_n_tiles, int num_cu, hipStream_t s) {
    (void)n;
    if (n_tiles <= 0) return hipSuccess;
    const int64_t units = (n_tiles + LB4_ROWS - 1) / LB4_ROWS;
    // split (one unit per workgroup) when the units could not even give every SIMD one wave; the grid must cover the
    // bound in that layout, which is only affordable for moderate bounds
    const bool may_split = d_n_tiles && units <= (int64_t)num_cu * 8 * 64;
    const int64_t split_units = may_split ? (int64_t)num_cu * 4 : -1;
    const int64_t blocks = may_split ? units : (units + 3) / 4;
    if (qd.normalised)
        hipLaunchKernelGGL((kvm_dtw_lb_tiled_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, series, prefix, qd, tiles,
                           n_tiles, out, counters, d_n_tiles, split_units);
    else
        hipLaunchKernelGGL((kvm_dtw_lb_tiled_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s, series, prefix, qd,
                           tiles, n_tiles, out, counters, d_n_tiles, split_units);
    return hipGetLastError();
}
