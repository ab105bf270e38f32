// Microbenchmark: how fast can ONE lane run a chain of dependent fp64 additions fed from LDS?  (The exact pass and the
// running-sum replay are exactly that.)  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off chain_bench.hip -o chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 8192
template <int MODE>
__global__ __launch_bounds__(256) void chain(const double *src, double *out, int reps) {
    __shared__ double buf[N];
    for (int i = threadIdx.x; i < N; i += blockDim.x) buf[i] = src[i];
    __syncthreads();
    double acc = 0.0, acc2 = 0.0;
    if (threadIdx.x == 0) {
        for (int r = 0; r < reps; r++) {
            if (MODE == 0) {  // one chain, 16 values per round, no prefetch
                for (int t = 0; t < N; t += 16) {
                    double2 v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const double2 *>(buf + t + 2 * u);
#pragma unroll
                    for (int u = 0; u < 8; u++) { acc += v[u].x; acc += v[u].y; }
                }
            } else if (MODE == 1) {  // one chain, prefetch next 16
                double2 a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; u++) a[u] = *reinterpret_cast<const double2 *>(buf + 2 * u);
                for (int t = 0; t < N; t += 32) {
#pragma unroll
                    for (int u = 0; u < 8; u++) b[u] = *reinterpret_cast<const double2 *>(buf + t + 16 + 2 * u);
#pragma unroll
                    for (int u = 0; u < 8; u++) { acc += a[u].x; acc += a[u].y; }
                    const int tn = (t + 32 < N) ? t + 32 : 0;
#pragma unroll
                    for (int u = 0; u < 8; u++) a[u] = *reinterpret_cast<const double2 *>(buf + tn + 2 * u);
#pragma unroll
                    for (int u = 0; u < 8; u++) { acc += b[u].x; acc += b[u].y; }
                }
            } else if (MODE == 2) {  // two independent chains (ex, ex2), prefetch
                double2 a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; u++) a[u] = *reinterpret_cast<const double2 *>(buf + 2 * u);
                for (int t = 0; t < N; t += 32) {
#pragma unroll
                    for (int u = 0; u < 8; u++) b[u] = *reinterpret_cast<const double2 *>(buf + t + 16 + 2 * u);
#pragma unroll
                    for (int u = 0; u < 8; u++) { acc += a[u].x; acc2 += a[u].y; }
                    const int tn = (t + 32 < N) ? t + 32 : 0;
#pragma unroll
                    for (int u = 0; u < 8; u++) a[u] = *reinterpret_cast<const double2 *>(buf + tn + 2 * u);
#pragma unroll
                    for (int u = 0; u < 8; u++) { acc += b[u].x; acc2 += b[u].y; }
                }
            } else if (MODE == 3) {  // pure register chain: no LDS at all (latency of a dependent v_add_f64)
                double v = src[0];
                for (int t = 0; t < N; t += 8) {
#pragma unroll
                    for (int u = 0; u < 8; u++) { acc += v; asm volatile("" : "+v"(acc)); }
                }
            }
        }
        out[0] = acc + acc2;
    }
    __syncthreads();  // (with 256 threads: three waves park here while lane 0 of wave 0 runs the chain)
}
int main() {
    std::vector<double> h(N);
    for (int i = 0; i < N; i++) h[i] = 1.0 / (i + 1);
    double *d, *o;
    hipMalloc(&d, N * 8); hipMalloc(&o, 8);
    hipMemcpy(d, h.data(), N * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 8;
    auto run = [&](auto kern, const char *name, int adds_per_elem) {
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, o, reps);
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, o, reps); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us per %d-element pass = %.2f ns per element\n", name, ms * 1e3 / reps, N, ms * 1e6 / reps / N);
    };
    run(chain<0>, "one chain, 16 per round, no prefetch", 1);
    run(chain<1>, "one chain, prefetch", 1);
    run(chain<2>, "two chains (x->acc, y->acc2), prefetch", 1);
    run(chain<3>, "register-only dependent adds", 1);
    {   // the same two-chain loop with three more waves of the workgroup parked at a barrier
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(chain<2>, dim3(1), dim3(256), 0, 0, d, o, reps);
        hipEventRecord(e0); hipLaunchKernelGGL(chain<2>, dim3(1), dim3(256), 0, 0, d, o, reps); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms2; hipEventElapsedTime(&ms2, e0, e1);
        printf("%-44s %8.1f us per pass = %.2f ns per element\n", "two chains, 256-thread workgroup", ms2 * 1e3 / reps, ms2 * 1e6 / reps / N);
    }
    // under load: the same with 1024 blocks resident
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(chain<1>, dim3(1024), dim3(64), 0, 0, d, o, reps);
    hipEventRecord(e0); hipLaunchKernelGGL(chain<1>, dim3(1024), dim3(64), 0, 0, d, o, reps); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("one chain, prefetch, 1024 blocks: %.2f ns per element\n", ms * 1e6 / reps / N);
    return 0;
}
