// Microbenchmark: what a chain of K short dependent kernels costs the HOST per call, from T threads on T streams --
// (a) K plain launches + a stream synchronisation, (b) one pre-instantiated graph whose K kernel nodes get new parameters
// (hipGraphExecKernelNodeSetParams) + one hipGraphLaunch, (c) stream capture of the K launches + hipGraphExecUpdate + launch.
// Build: hipcc --offload-arch=gfx950 -O2 launch_chain.hip -o launch_chain -lpthread     Run: ./launch_chain [K] [calls]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct big_arg { double v[24]; };  // (the library's kernels take ~200-byte structs by value)
__global__ void step_kernel(unsigned long long *c, int n, big_arg a) {
    if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += (unsigned long long)n + (unsigned long long)a.v[3];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void worker(int mode, int K, int calls, double *out_us) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *c; CK(hipMalloc(&c, 64)); CK(hipMemset(c, 0, 64));
    big_arg a{}; a.v[3] = 1.0;
    hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr; std::vector<hipGraphNode_t> nodes;
    if (mode == 1) {
        CK(hipGraphCreate(&g, 0));
        hipGraphNode_t prev = nullptr;
        for (int k = 0; k < K; k++) {
            int n = 1; void *args[3] = {&c, &n, &a};
            hipKernelNodeParams p{}; p.func = (void *)step_kernel; p.gridDim = dim3(64); p.blockDim = dim3(256); p.kernelParams = args;
            hipGraphNode_t nd; CK(hipGraphAddKernelNode(&nd, g, prev ? &prev : nullptr, prev ? 1 : 0, &p)); nodes.push_back(nd); prev = nd;
        }
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    }
    double t0 = 0;
    for (int i = -20; i < calls; i++) {
        if (i == 0) t0 = now();
        if (mode == 0) {
            for (int k = 0; k < K; k++) hipLaunchKernelGGL(step_kernel, dim3(64 + (i & 7)), dim3(256), 0, s, c, i + k, a);
        } else if (mode == 1) {
            for (int k = 0; k < K; k++) {
                int n = i + k; void *args[3] = {&c, &n, &a};
                hipKernelNodeParams p{}; p.func = (void *)step_kernel; p.gridDim = dim3(64 + (i & 7)); p.blockDim = dim3(256); p.kernelParams = args;
                CK(hipGraphExecKernelNodeSetParams(ex, nodes[k], &p));
            }
            CK(hipGraphLaunch(ex, s));
        } else {
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < K; k++) hipLaunchKernelGGL(step_kernel, dim3(64 + (i & 7)), dim3(256), 0, s, c, i + k, a);
            hipGraph_t cg; CK(hipStreamEndCapture(s, &cg));
            if (!ex) CK(hipGraphInstantiate(&ex, cg, nullptr, nullptr, 0));
            else {
                hipGraphNode_t bad; hipGraphExecUpdateResult res;
                if (hipGraphExecUpdate(ex, cg, &bad, &res) != hipSuccess) { CK(hipGraphExecDestroy(ex)); CK(hipGraphInstantiate(&ex, cg, nullptr, nullptr, 0)); }
            }
            CK(hipGraphLaunch(ex, s));
            CK(hipGraphDestroy(cg));
        }
        CK(hipStreamSynchronize(s));
    }
    *out_us = (now() - t0) / calls * 1e6;
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 7, calls = argc > 2 ? atoi(argv[2]) : 2000;
    const char *names[3] = {"plain launches", "graph + SetParams", "capture + ExecUpdate"};
    for (int mode = 0; mode < 3; mode++)
        for (int T : {1, 2, 5}) {
            std::vector<double> us(T); std::vector<std::thread> th;
            const double t0 = now();
            for (int t = 0; t < T; t++) th.emplace_back(worker, mode, K, calls, &us[t]);
            for (auto &x : th) x.join();
            const double wall = now() - t0;
            printf("%-22s K=%d threads=%d: %.1f us per call per thread, %.1f us per call overall\n", names[mode], K, T, us[0], wall / (calls * T) * 1e6);
            fflush(stdout);
        }
    return 0;
}
