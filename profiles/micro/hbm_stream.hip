// hbm_stream.hip -- which streaming kernel shape reads the HBM rate the guide quotes (6.0 - 6.3 TB/s)?  copy / triad / read-only,
// 16 B per lane, U loads in flight per lane, plain or non-temporal; 3 x 1 GiB arrays.   hipcc --offload-arch=gfx950 -O3 -o hbm_stream hbm_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double double2_ __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_k(double2_ *__restrict__ a, const double2_ *__restrict__ b, int64_t n2) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n2; i += U * stride) {
        double2_ v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(b + i + u * stride) : b[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], a + i + u * stride); else a[i + u * stride] = v[u]; }
    }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void triad_k(double2_ *__restrict__ a, const double2_ *__restrict__ b, const double2_ *__restrict__ c, double s, int64_t n2) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n2; i += U * stride) {
        double2_ x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; u++) { x[u] = NT ? __builtin_nontemporal_load(b + i + u * stride) : b[i + u * stride]; y[u] = NT ? __builtin_nontemporal_load(c + i + u * stride) : c[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; u++) { double2_ r = x[u] + s * y[u]; if (NT) __builtin_nontemporal_store(r, a + i + u * stride); else a[i + u * stride] = r; }
    }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_k(double *out, const double2_ *__restrict__ b, int64_t n2) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double2_ acc = {0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n2; i += U * stride) {
        double2_ v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(b + i + u * stride) : b[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc.x + acc.y == 12345.678) out[0] = acc.x;
}
int main() {
    const int64_t n = (int64_t)128 << 20;  // doubles per array = 1 GiB
    double *a; hipMalloc((void **)&a, 3 * n * 8); hipMemset(a, 0, 3 * n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch, double bytes) {
        for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
            launch(blocks); hipEventRecord(e0, 0);
            for (int r = 0; r < 6; r++) launch(blocks);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-16s blocks %5d  %7.1f GB/s\n", name, blocks, 6 * bytes / (ms * 1e-3) / 1e9);
        }
    };
    auto *A = (double2_ *)a; auto *B = (double2_ *)(a + n); auto *C = (double2_ *)(a + 2 * n);
    const int64_t n2 = n / 2;
#define L(K, ...) [&](int bl) { hipLaunchKernelGGL(K, dim3(bl), dim3(256), 0, 0, __VA_ARGS__); }
    run("copy U1", L((copy_k<1, false>), A, B, n2), 16.0 * n);
    run("copy U4", L((copy_k<4, false>), A, B, n2), 16.0 * n);
    run("copy U4 nt", L((copy_k<4, true>), A, B, n2), 16.0 * n);
    run("triad U1", L((triad_k<1, false>), A, B, C, 1.5, n2), 24.0 * n);
    run("triad U2", L((triad_k<2, false>), A, B, C, 1.5, n2), 24.0 * n);
    run("triad U4", L((triad_k<4, false>), A, B, C, 1.5, n2), 24.0 * n);
    run("triad U4 nt", L((triad_k<4, true>), A, B, C, 1.5, n2), 24.0 * n);
    run("read U1", L((read_k<1, false>), a, B, n2), 8.0 * n);
    run("read U4", L((read_k<4, false>), a, B, n2), 8.0 * n);
    run("read U4 nt", L((read_k<4, true>), a, B, n2), 8.0 * n);
    return 0;
}
