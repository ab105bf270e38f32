// What does a workgroup that finds nothing cost when it needs 63 KB of LDS?  The ED pruner's launches over device-built lists
// size their grid for the bounds (3 088 workgroups of 256 threads, 64 512 B of LDS each, for the bench list) and nearly all of
// them return at once.  This times back-to-back launches of such a kernel, on one stream and on eight, for grids of 8 .. 3 088
// workgroups, with and without the LDS.      hipcc --offload-arch=gfx950 -O2 empty_lds_grid.hip -o /tmp/empty_lds_grid
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
template <int LDS_DOUBLES>
__global__ __launch_bounds__(256, 2) void k_empty(int *p) {
    __shared__ double l[LDS_DOUBLES];
    if (p[0] == 12345) {  // never: keeps the allocation
        l[threadIdx.x] = p[1];
        __syncthreads();
        p[2] = (int)l[(threadIdx.x + 1) & 255];
    }
}
template <int L>
static double run(int grid, int streams, int n, int *d) {
    std::vector<hipStream_t> st(streams);
    for (auto &s : st) hipStreamCreate(&s);
    for (auto &s : st) hipLaunchKernelGGL(k_empty<L>, dim3(grid), dim3(256), 0, s, d);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < streams; t++)
        th.emplace_back([&, t] {
            hipSetDevice(0);
            for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty<L>, dim3(grid), dim3(256), 0, st[t], d);
            hipStreamSynchronize(st[t]);
        });
    for (auto &x : th) x.join();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    for (auto &s : st) hipStreamDestroy(s);
    return us / ((double)n * streams);
}
int main() {
    int *d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    const int grids[] = {8, 512, 1024, 3088};
    for (int streams : {1, 8})
        for (int g : grids)
            printf("streams %d grid %4d: %6.2f us per launch with 64512 B of LDS, %6.2f with 2048 B\n", streams, g,
                   run<8064>(g, streams, 2000, d), run<256>(g, streams, 2000, d));
    return 0;
}
