// kvm_selftest.hip -- what THIS device sustains, measured by the caller's own run (kvm_measure_device): the roofline lines
// of bench.py quote vendor peaks; these three numbers say what the box under the run delivers.
//   0  HBM read stream    16 B per lane, 1 GiB swept in order (beyond the 256 MB Infinity Cache): GB/s.  What the ED pruner does
//                         to HBM is READ windows; a triad (two reads, one write: rounds 4-5) reads 4.6 - 5.4 TB/s on the box
//                         where this reads 6.4 - 6.5 and the guide's float4 copy 6.3 (profiles/micro/hbm_stream.hip,
//                         profiles/r06_hbm_stream.log)
//   1  fp64 FMA issue     register-only v_fma_f64, 8 independent chains per lane, 4 waves per SIMD: TFLOP/s (FMA = 2)
//   2  fp64 non-FMA issue the banded DP's cell mix (sub, mul, min, min, add), register-only: Tops/s
#include "kvm_internal.h"

__global__ __launch_bounds__(256) void kvm_read_stream_kernel(double *__restrict__ out, const double *__restrict__ b, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 2;
    double2 acc = make_double2(0.0, 0.0);
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i + 1 < n; i += stride) {
        const double2 x = *reinterpret_cast<const double2 *>(b + i);
        acc.x += x.x;
        acc.y += x.y;
    }
    if (acc.x + acc.y == 12345.678) out[0] = acc.x;  // (never true: keeps the loads alive)
}

__global__ __launch_bounds__(256) void kvm_fma_rate_kernel(double *out, double x, double y, int iters) {
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; u++) acc[u] = x * (u + 1) + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) acc[u] = __builtin_fma(acc[u], y, x);
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) s += acc[u];
    if (s == 12345.678) out[0] = s;  // (never true: keeps the chains alive)
}

__global__ __launch_bounds__(256) void kvm_cellmix_rate_kernel(double *out, double x, double y, int iters) {
    // eight independent "cells": v = min(min(v, p), q) + (a - b) * (a - b), written without contraction
    double v[8], a[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { v[u] = x * (u + 1); a[u] = y + threadIdx.x * 1e-3 + u; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const double d = a[u] - v[(u + 1) & 7];
            const double dd = d * d;
            double mn;
            asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(v[u]), "v"(v[(u + 3) & 7]));
            asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(mn), "v"(v[(u + 5) & 7]));
            v[u] = mn + dd;
        }
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) s += v[u];
    if (s == 12345.678) out[0] = s;
}

#define ST_CHK(x)                                                   \
    do {                                                            \
        const hipError_t e_ = (x);                                  \
        if (e_ != hipSuccess) {                                     \
            if (a) (void)hipFree(a);                                \
            return e_;                                              \
        }                                                           \
    } while (0)

hipError_t kvm_measure_device_rate(int what, int num_cu, hipStream_t s, hipEvent_t e0, hipEvent_t e1, double *value) {
    double *a = nullptr;
    float ms = 0;
    if (what == 0) {
        const int64_t n = (int64_t)128 << 20;  // 128 Mi doubles = 1 GiB
        ST_CHK(hipMalloc((void **)&a, (size_t)n * sizeof(double) + 64));
        ST_CHK(hipMemsetAsync(a, 0, (size_t)n * sizeof(double) + 64, s));
        const int reps = 8;
        hipLaunchKernelGGL(kvm_read_stream_kernel, dim3(num_cu * 8), dim3(256), 0, s, a + n, a, n);
        ST_CHK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kvm_read_stream_kernel, dim3(num_cu * 8), dim3(256), 0, s, a + n, a, n);
        ST_CHK(hipEventRecord(e1, s));
        ST_CHK(hipEventSynchronize(e1));
        ST_CHK(hipEventElapsedTime(&ms, e0, e1));
        *value = (double)reps * 8.0 * (double)n / (ms * 1e-3) / 1e9;
        (void)hipFree(a);
        return hipGetLastError();
    }
    ST_CHK(hipMalloc((void **)&a, 64));
    const int iters = 1 << 15, blocks = num_cu * 4;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    for (int pass = 0; pass < 2; pass++) {  // (the first pass brings the clocks up)
        ST_CHK(hipEventRecord(e0, s));
        if (what == 1) hipLaunchKernelGGL(kvm_fma_rate_kernel, dim3(blocks), dim3(256), 0, s, a, 0.999, 1.0000001, iters);
        else hipLaunchKernelGGL(kvm_cellmix_rate_kernel, dim3(blocks), dim3(256), 0, s, a, 0.999, 1.5, iters);
        ST_CHK(hipEventRecord(e1, s));
        ST_CHK(hipEventSynchronize(e1));
        ST_CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double lanes = (double)blocks * 256.0, per_lane = (double)iters * 8.0 * (what == 1 ? 2.0 : 5.0);
    *value = lanes * per_lane / (ms * 1e-3) / 1e12;
    (void)hipFree(a);
    return hipGetLastError();
}
