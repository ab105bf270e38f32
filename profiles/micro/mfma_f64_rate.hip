// Microbenchmark: sustained rate of v_mfma_f64_16x16x4_f64 on gfx950, alone and fed from LDS (2 ds_read_b64 per MFMA),
// next to the vector v_fma_f64 rate, at 1 / 2 / 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate
// mfma_f64_rate.hip ; prints TFLOP/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 3 (round 4): do the matrix pipe and the vector pipe add up?  Waves 0 and 2 of every workgroup run
// the MFMA loop, waves 1 and 3 the v_fma_f64 loop: with 4 workgroup-waves per SIMD every SIMD holds two of each kind.
template <int MODE>  // 0: MFMA from registers, 1: MFMA with operands from LDS, 2: vector FMA (16 per step, 8 accumulators x2)
__global__ __launch_bounds__(256) void rate_kernel(double *out, int iters) {
    __shared__ double lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    d4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    double a = 1.0 + lane * 1e-6, b = 1.0 - lane * 1e-6;
    double v[16];
    for (int k = 0; k < 16; k++) v[k] = k * 1e-3;
    const double *p = lds + (threadIdx.x >> 6) * 1024 + lane;
    const bool mfma_wave = MODE == 3 && (((threadIdx.x >> 6) & 1) == 0);
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || (MODE == 3 && mfma_wave)) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, acc3, 0, 0, 0);
        } else if (MODE == 1) {
            const int o = (it & 7) * 64;
            const double x0 = p[o], x1 = p[o + 512 - 448], x2 = p[(o + 128) & 511], x3 = p[(o + 192) & 511], q = p[(o + 256) & 511];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, q, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, q, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, q, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, q, acc3, 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = __builtin_fma(v[k], a, b);
        }
    }
    double s = 0;
    for (int k = 0; k < 4; k++) s += acc0[k] + acc1[k] + acc2[k] + acc3[k];
    for (int k = 0; k < 16; k++) s += v[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, int blocks_per_cu, int num_cu) {
    const int blocks = blocks_per_cu * num_cu, iters = 20000;
    double *out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per iteration per wave: MODE 0/1: 4 MFMA x 2048 flop; MODE 2: 64 FMA instr x 64 lanes x 2 flop; MODE 3: half the waves each
    const double flop = (double)blocks * 4 * iters * (MODE == 2 ? 64.0 * 64 * 2 : MODE == 3 ? 0.5 * (64.0 * 64 * 2 + 4.0 * 2048) : 4.0 * 2048);
    printf("%-28s %d waves/SIMD: %8.3f ms  %7.2f TFLOP/s", name, blocks_per_cu, ms, flop / ms / 1e9);
    if (MODE == 3) printf("  (vector part %.2f + matrix part %.2f)", (double)blocks * 2 * iters * 64.0 * 64 * 2 / ms / 1e9, (double)blocks * 2 * iters * 4.0 * 2048 / ms / 1e9);
    printf("\n");
    hipFree(out);
}

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cu = pr.multiProcessorCount;
    printf("%s, %d CUs\n", pr.gcnArchName, cu);
    for (int w : {1, 2, 4}) {
        run<0>("mfma_f64 16x16x4 (regs)", w, cu);
        run<1>("mfma_f64 16x16x4 (LDS fed)", w, cu);
        run<2>("v_fma_f64", w, cu);
        if (w >= 2) run<3>("half mfma_f64 + half v_fma", w, cu);
    }
    return 0;
}
