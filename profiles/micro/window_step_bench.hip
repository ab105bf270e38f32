// Microbenchmark: cost of one WINDOW STEP of the running-sum replay (a += d; record a; a -= o) on one lane, by variant.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off window_step_bench.hip -o window_step_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 4096
#define FENCE2(a, b) asm volatile("" : "+v"(a), "+v"(b)::"memory")
// MODE 0: LDS reads (R = 4 per register set, one round ahead) + one 8-byte global store per step   [the kernels today]
// MODE 1: the same without the store
// MODE 2: one 16-byte global store per two steps
// MODE 3: stores only (operands are register constants)
// MODE 4: record into LDS (8-byte ds_write per step) instead of global memory
// MODE 5: one 16-byte LDS write per two steps
// MODE 6: operands straight from global memory at uniform addresses (scalar loads, no LDS) + 8-byte global store per step
template <int MODE>
__global__ __launch_bounds__(256) void steps(const double *__restrict__ src, double *__restrict__ out, int reps) {
    __shared__ __attribute__((aligned(16))) double dbuf[N], obuf[N], rec[N];
    for (int i = threadIdx.x; i < N; i += blockDim.x) { dbuf[i] = src[i]; obuf[i] = src[N - 1 - i]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a_l = 0.0, dummy = 0.0;
        double *o = out + (size_t)blockIdx.x * N;
        for (int r = 0; r < reps; r++) {
            constexpr int R = 4;
            double dv[2][R], ov[2][R];
            auto fetch = [&](int t0, int set) {
#pragma unroll
                for (int u = 0; u < R; u++) {
                    if (MODE == 3) { dv[set][u] = 1.0; ov[set][u] = 0.5; }
                    else if (MODE == 6) { dv[set][u] = src[t0 + u]; ov[set][u] = src[t0 + u + 37]; }
                    else { dv[set][u] = dbuf[t0 + u]; ov[set][u] = obuf[t0 + u]; }
                }
            };
            auto consume = [&](int t0, int set) {
                if (MODE == 2 || MODE == 5) {
#pragma unroll
                    for (int u = 0; u < R; u += 2) {
                        double2 w;
                        a_l += dv[set][u]; w.x = a_l; a_l -= ov[set][u];
                        a_l += dv[set][u + 1]; w.y = a_l; a_l -= ov[set][u + 1];
                        if (MODE == 2) *reinterpret_cast<double2 *>(o + t0 + u) = w;
                        else *reinterpret_cast<double2 *>(rec + t0 + u) = w;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < R; u++) {
                        a_l += dv[set][u];
                        if (MODE == 0 || MODE == 3 || MODE == 6) o[t0 + u] = a_l;
                        if (MODE == 4) rec[t0 + u] = a_l;
                        a_l -= ov[set][u];
                    }
                }
            };
            const int rounds = N / (2 * R);
            fetch(0, 0);
            for (int rr = 0; rr < rounds; rr++) {
                const int t = rr * 2 * R;
                fetch(t + R, 1);
                FENCE2(a_l, dummy);
                consume(t, 0);
                fetch((rr + 1 < rounds) ? t + 2 * R : 0, 0);
                FENCE2(a_l, dummy);
                consume(t + R, 1);
            }
        }
        o[0] = a_l + rec[5];
    }
    __syncthreads();
}
int main() {
    std::vector<double> h(N);
    for (int i = 0; i < N; i++) h[i] = 1.0 / (i + 1);
    double *d, *o;
    hipMalloc(&d, (N + 64) * 8); hipMalloc(&o, (size_t)2048 * N * 8);
    hipMemcpy(d, h.data(), N * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 8;
    auto run = [&](auto kern, const char *name, int blocks) {
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, o, reps);
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, o, reps); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s blocks=%4d  %.2f ns per step\n", name, blocks, ms * 1e6 / reps / N);
    };
    for (int blocks : {1, 1024}) {
        run(steps<0>, "LDS reads + 8-byte global store per step", blocks);
        run(steps<1>, "LDS reads, no store", blocks);
        run(steps<2>, "LDS reads + 16-byte global store per 2 steps", blocks);
        run(steps<3>, "register operands + 8-byte global store per step", blocks);
        run(steps<4>, "LDS reads + 8-byte LDS write per step", blocks);
        run(steps<5>, "LDS reads + 16-byte LDS write per 2 steps", blocks);
        run(steps<6>, "scalar loads from global + 8-byte global store per step", blocks);
    }
    return 0;
}
