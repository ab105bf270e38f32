// kvm_dtw_ref.hip -- the sequential, reference-ordered path of the DTW cascade for gfx950.
//
// A candidate lands here when one of the quantities the reference compares with eps^2 came out within rounding of eps^2
// in the wave-parallel arithmetic of the DP kernel's prologue / row test (kvm_dtw.hip) -- on continuous data practically
// never, on integer-valued data (where sums are exact and ties with an integer eps^2 are common) often, and always for
// eps = 0.  Here the cascade in front of the DP is evaluated literally (K/NormQueryEngineDtw.java:555-587,
// K/QueryEngineDtw.java:411-441), one lane per candidate, every addition in the reference's order:
//   lbKimHierarchy with its early returns (K/utils/DtwUtils.java:149-189)                      -> `lbKim <= eps^2`
//   lbKeoghCumulative: `for (i = 0; i < len && lb < bsf; i++)`, cb1[order[i]] = d (:206-222)   -> `lbK <= eps^2`
//   lbKeoghDataCumulative on the chunk-clipped data envelope, cb2[] (:238-257, :50-91)          -> `lbK2 <= eps^2`
//   cb[] = suffix sums of cb1 if lbK > lbK2 else of cb2, added from the back (K/NormQueryEngineDtw.java:576-587)
// A loop that stops early leaves the rest of its cb array at zero, exactly as `new double[queryLength]` does.  The DP
// itself then runs once more in the DP kernel, which reads these cb[] arrays and applies the reference's row test
// `min_cost + cb[i+r+1] >= bsf` without any tolerance (every DP cell is bit-exact by construction).
// With the replayed mean/std these are the reference's own values bit for bit (no FMA: -ffp-contract=off).
#include "kvm_dtw_ref.h"

// upper / lower envelope of the fetched chunk at 1-based point p: max / min of data[max(cbeg,p-r) .. min(cend,p+r)]
// (what lowerUpperLemire computes, a pure selection); interior points come from the whole-series envelope
__device__ __forceinline__ void ref_env_at(const double *__restrict__ series, const kvm_dtw_aux &ax, int r, int32_t cbeg,
                                           int32_t cend, int32_t p, double &U, double &L) {
    if (p - r >= cbeg && p + r <= cend) {
        U = ax.env_u[p - 1];
        L = ax.env_l[p - 1];
        return;
    }
    const int32_t a = max(cbeg, p - r), b = min(cend, p + r);
    double mx = series[a - 1], mn = mx;
    for (int32_t t = a + 1; t <= b; t++) {
        const double v = series[t - 1];
        mx = (v > mx) ? v : mx;
        mn = (v < mn) ? v : mn;
    }
    U = mx;
    L = mn;
}

__global__ __launch_bounds__(64) void kvm_dtw_ref_chains_kernel(const double *__restrict__ series, kvm_query_dev qd,
                                                                kvm_dtw_aux ax, kvm_cand *__restrict__ cands, int64_t n,
                                                                double *__restrict__ slots, int keep_slots) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    kvm_cand c = cands[idx];
    if (c.start < 0) return;
    const int m = qd.m, r = qd.rho;
    const double bsf = qd.eps2, mean = c.mean, std = c.std;
    const double *__restrict__ x = series + (c.start - 1);
    double *__restrict__ cb1 = slots + (size_t)idx * 2 * m, *__restrict__ cb2 = cb1 + m;
    bool pass = true;
    if (m >= 6) {
        const double lbKim = dref_kim(x, qd.q, m, mean, std, bsf);
        pass = lbKim <= bsf;                                                   // K/NormQueryEngineDtw.java:556
    }
    double lbK = 0, lbK2 = 0;
    if (pass) {
        if (!keep_slots)
            for (int k = 0; k < m; k++) cb1[k] = 0.0;                          // new double[queryLength]
        for (int i = 0; i < m && lbK < bsf; i++) {                             // K/utils/DtwUtils.java:210
            const int o = ax.order[i];
            const double d = dref_term_q((x[o] - mean) / std, ax.uo[i], ax.lo[i]);
            lbK += d;
            cb1[o] = d;
        }
        pass = lbK <= bsf;                                                     // :561
    }
    if (pass) {
        int32_t cbeg, cend;
        dref_chunk(ax, m, c.run, cbeg, cend);
        if (!keep_slots)
            for (int k = 0; k < m; k++) cb2[k] = 0.0;
        for (int i = 0; i < m && lbK2 < bsf; i++) {                            // K/utils/DtwUtils.java:242
            const int o = ax.order[i];
            double U, L;
            ref_env_at(series, ax, r, cbeg, cend, c.start + o, U, L);
            const double d = dref_term_d(ax.qo[i], U, L, mean, std);
            lbK2 += d;
            cb2[o] = d;
        }
        pass = lbK2 <= bsf;                                                    // K/NormQueryEngineDtw.java:573
    }
    if (pass) {
        double *__restrict__ cb = (lbK > lbK2) ? cb1 : cb2;                    // :577
        for (int k = m - 2; k >= 0; k--) cb[k] = cb[k + 1] + cb[k];            // :578-586, in place
        c.lim = (lbK > lbK2) ? 0.0 : 1.0;  // which half of the slot holds cb[] (read by the DP kernel)
    } else {
        c.start = -c.start;
    }
    cands[idx] = c;
}

hipError_t kvm_launch_dtw_ref_chains(const double *series, const kvm_query_dev &qd, const kvm_dtw_aux &ax, kvm_cand *cands,
                                     int64_t n, double *slots, hipStream_t s, bool keep_slots) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(kvm_dtw_ref_chains_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, series, qd, ax, cands, n,
                       slots, keep_slots ? 1 : 0);
    return hipGetLastError();
}
