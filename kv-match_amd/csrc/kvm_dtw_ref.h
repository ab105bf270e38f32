// kvm_dtw_ref.h -- device helpers shared by the DP kernels (kvm_dtw.hip) and the sequential reference-ordered path
// (kvm_dtw_ref.hip): the pieces of the reference's DTW cascade that must be evaluated exactly as written, because the
// reference compares their results with eps^2 by `>=` / `<=` and a tie changes the answer set (see kvm_internal.h,
// "DTW cascade").  Everything here runs with the replayed (reference-exact) mean / std of the candidate; no FMA
// (-ffp-contract=off), true divisions, the reference's comparison forms.
#pragma once
#include "kvm_internal.h"

__device__ __forceinline__ double dref_dsq(double a, double b) { return (a - b) * (a - b); }        // DtwUtils.dist, :38-40
__device__ __forceinline__ double dref_min(double a, double b) { return (a < b) ? a : b; }          // DtwUtils.min, :34-36

// lbKimHierarchy, K/utils/DtwUtils.java:149-189, with its early returns; x = first point of the window (the reference
// indexes a circular buffer, t[j + k] is the window's k-th point).
__device__ __forceinline__ double dref_kim(const double *__restrict__ x, const double *__restrict__ q, int len, double mean,
                                           double std, double bsf) {
    double d, lb;
    const double x0 = (x[0] - mean) / std;
    const double y0 = (x[len - 1] - mean) / std;
    lb = dref_dsq(x0, q[0]) + dref_dsq(y0, q[len - 1]);
    if (lb >= bsf) return lb;
    const double x1 = (x[1] - mean) / std;
    d = dref_min(dref_dsq(x1, q[0]), dref_dsq(x0, q[1]));
    d = dref_min(d, dref_dsq(x1, q[1]));
    lb += d;
    if (lb >= bsf) return lb;
    const double y1 = (x[len - 2] - mean) / std;
    d = dref_min(dref_dsq(y1, q[len - 1]), dref_dsq(y0, q[len - 2]));
    d = dref_min(d, dref_dsq(y1, q[len - 2]));
    lb += d;
    if (lb >= bsf) return lb;
    const double x2 = (x[2] - mean) / std;
    d = dref_min(dref_dsq(x0, q[2]), dref_dsq(x1, q[2]));
    d = dref_min(d, dref_dsq(x2, q[2]));
    d = dref_min(d, dref_dsq(x2, q[1]));
    d = dref_min(d, dref_dsq(x2, q[0]));
    lb += d;
    if (lb >= bsf) return lb;
    const double y2 = (x[len - 3] - mean) / std;
    d = dref_min(dref_dsq(y0, q[len - 3]), dref_dsq(y1, q[len - 3]));
    d = dref_min(d, dref_dsq(y2, q[len - 3]));
    d = dref_min(d, dref_dsq(y2, q[len - 2]));
    d = dref_min(d, dref_dsq(y2, q[len - 1]));
    lb += d;
    return lb;
}

// one term of lbKeoghCumulative (:210-219): xz = (t - mean) / std against the query envelope at the same sorted index
__device__ __forceinline__ double dref_term_q(double xz, double u, double l) {
    double d = 0;
    if (xz > u) d = dref_dsq(xz, u);
    else if (xz < l) d = dref_dsq(xz, l);
    return d;
}
// one term of lbKeoghDataCumulative (:242-254): query point against the normalised data envelope
__device__ __forceinline__ double dref_term_d(double qv, double U, double L, double mean, double std) {
    const double uu = (U - mean) / std, ll = (L - mean) / std;
    double d = 0;
    if (qv > uu) d = dref_dsq(qv, uu);
    else if (qv < ll) d = dref_dsq(qv, ll);
    return d;
}

// Data range [cbeg, cend] (1-based points) the reference fetched for the candidate's interval -- where its data envelope
// (lowerUpperLemire over the fetched chunk, K/QueryEngineDtw.java:397-399, K/NormQueryEngineDtw.java:522-524) is clipped.
// RSM engines fetch per interval: begin = the run's first window, end = its last window's last point
// (K/QueryEngineDtw.java:387-394); cNSM engines fetch merged chunks, whose bounds the host computed per run.
__device__ __forceinline__ void dref_chunk(const kvm_dtw_aux &ax, int m, int32_t run, int32_t &cbeg, int32_t &cend) {
    const int64_t r = run;
    if (ax.run_chunk) {
        const int2 c = ax.run_chunk[r];
        cbeg = c.x;
        cend = c.y;
    } else {
        const kvm_run run = ax.runs[r];
        cbeg = run.start;
        cend = run.start + run.count + m - 2;
    }
}
