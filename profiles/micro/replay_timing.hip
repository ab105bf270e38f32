// Diagnostics: cycle stamps inside kvm_replay_runs_kernel for one run of one window (m = 8192): where does a chunk's
// time go -- the sequential consumer lane or the producers?  Single translation unit with the library's kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DKVM_EX_TIMING -I../../include replay_timing.hip -o replay_timing
#include "../../kv-match_amd/csrc/kvm_exact.hip"
#include <cstdio>
#include <vector>
int main() {
    const int m = 8192, n = 40000;
    std::vector<double> h(n);
    for (int i = 0; i < n; i++) h[i] = std::sin(i * 0.01) + 1e-3 * (i % 17);
    double *x; hipMalloc(&x, n * 8); hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    kvm_run run{1, 1, 0, -1, 0};
    kvm_run *d_runs; hipMalloc(&d_runs, sizeof run); hipMemcpy(d_runs, &run, sizeof run, hipMemcpyHostToDevice);
    int32_t one = 1, zero = 0, *maxk, *list; hipMalloc(&maxk, 4); hipMalloc(&list, 4);
    hipMemcpy(maxk, &one, 4, hipMemcpyHostToDevice); hipMemcpy(list, &zero, 4, hipMemcpyHostToDevice);
    unsigned long long n1 = 1, *d_n; hipMalloc(&d_n, 8); hipMemcpy(d_n, &n1, 8, hipMemcpyHostToDevice);
    double2 *stats; hipMalloc(&stats, 1024);
    kvm_replay_ws ws{d_runs, 1, maxk, stats, list, d_n};
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(kvm_replay_runs_kernel, dim3(1), dim3(EX_THREADS), 0, 0, x, m, ws);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> dbg(8192);
    hipMemcpyFromSymbol(dbg.data(), HIP_SYMBOL(kvm_dbg), 8192 * 8);
    const unsigned long long t0 = dbg[0];
    for (int c = 0; c < 16; c++)
        printf("chunk %2d: consume %6lld .. %6lld (%5lld cyc)   produce(next) %6lld .. %6lld (%5lld cyc)\n", c,
               (long long)(dbg[4 * c] - t0), (long long)(dbg[4 * c + 1] - t0), (long long)(dbg[4 * c + 1] - dbg[4 * c]),
               (long long)(dbg[4 * c + 2] - t0), (long long)(dbg[4 * c + 3] - t0), (long long)(dbg[4 * c + 3] - dbg[4 * c + 2]));
    return 0;
}
