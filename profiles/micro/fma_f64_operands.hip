// Microbenchmark: does the VGPR placement of the three 64-bit operands of v_fmac_f64 (acc += w * q) change its issue rate
// on gfx950?  Eight accumulators, eight window registers, one query register, all in fixed registers: every operand's
// register index is 0 mod 4 ("class 0") or 2 mod 4 ("class 1").  Variants: which class w and q are in relative to acc.
// In-kernel stamps (s_memtime / s_memrealtime around the loop of one wave) give cycles per instruction and the clock.
// Build: hipcc --offload-arch=gfx950 -O3 -o fma_f64_operands fma_f64_operands.hip ; prints TFLOP/s per variant at 1 / 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

// acc: v[0:1], v[4:5], ... v[28:29] (class 0).  w: v[32+WO+4k : ...], q: v[68+QO : ...]; WO, QO in {0, 2}
#define FM(a, w, q) "v_fmac_f64 v[" #a ":" #a "+1], v[" #w ":" #w "+1], v[" #q ":" #q "+1]\n"
#define BLOCK8(W0, W1, W2, W3, W4, W5, W6, W7, Q) \
    FM(0, W0, Q) FM(4, W1, Q) FM(8, W2, Q) FM(12, W3, Q) FM(16, W4, Q) FM(20, W5, Q) FM(24, W6, Q) FM(28, W7, Q)
#define CLOB "v0","v1","v4","v5","v8","v9","v12","v13","v16","v17","v20","v21","v24","v25","v28","v29", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51", \
             "v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v68","v69","v70","v71"

template <int VAR>
__global__ __launch_bounds__(256) void k(double *out, int iters, unsigned long long *stamps) {
    // initialise every register the blocks read
    asm volatile("v_mov_b32 v0, 0\nv_mov_b32 v1, 0\nv_mov_b32 v4, 0\nv_mov_b32 v5, 0\nv_mov_b32 v8, 0\nv_mov_b32 v9, 0\nv_mov_b32 v12, 0\nv_mov_b32 v13, 0\n"
                 "v_mov_b32 v16, 0\nv_mov_b32 v17, 0\nv_mov_b32 v20, 0\nv_mov_b32 v21, 0\nv_mov_b32 v24, 0\nv_mov_b32 v25, 0\nv_mov_b32 v28, 0\nv_mov_b32 v29, 0\n" ::: CLOB);
#pragma unroll
    for (int r = 32; r < 72; r++) {}
    asm volatile("v_mov_b32 v32, 0\nv_mov_b32 v33, 0x3ff00000\nv_mov_b32 v34, 0\nv_mov_b32 v35, 0x3ff00000\nv_mov_b32 v36, 0\nv_mov_b32 v37, 0x3ff00000\nv_mov_b32 v38, 0\nv_mov_b32 v39, 0x3ff00000\n"
                 "v_mov_b32 v40, 0\nv_mov_b32 v41, 0x3ff00000\nv_mov_b32 v42, 0\nv_mov_b32 v43, 0x3ff00000\nv_mov_b32 v44, 0\nv_mov_b32 v45, 0x3ff00000\nv_mov_b32 v46, 0\nv_mov_b32 v47, 0x3ff00000\n"
                 "v_mov_b32 v48, 0\nv_mov_b32 v49, 0x3ff00000\nv_mov_b32 v50, 0\nv_mov_b32 v51, 0x3ff00000\nv_mov_b32 v52, 0\nv_mov_b32 v53, 0x3ff00000\nv_mov_b32 v54, 0\nv_mov_b32 v55, 0x3ff00000\n"
                 "v_mov_b32 v56, 0\nv_mov_b32 v57, 0x3ff00000\nv_mov_b32 v58, 0\nv_mov_b32 v59, 0x3ff00000\nv_mov_b32 v60, 0\nv_mov_b32 v61, 0x3ff00000\nv_mov_b32 v62, 0\nv_mov_b32 v63, 0x3ff00000\n"
                 "v_mov_b32 v64, 0\nv_mov_b32 v65, 0x3ff00000\nv_mov_b32 v68, 0\nv_mov_b32 v69, 0x3e700000\nv_mov_b32 v70, 0\nv_mov_b32 v71, 0x3e700000\n" ::: CLOB);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (VAR == 0)       // w class 0, q class 0: all three operands in the same class
            asm volatile(BLOCK8(32, 36, 40, 44, 48, 52, 56, 60, 68) BLOCK8(36, 40, 44, 48, 52, 56, 60, 64, 68) ::: CLOB);
        else if (VAR == 1)  // w class 1, q class 0
            asm volatile(BLOCK8(34, 38, 42, 46, 50, 54, 58, 62, 68) BLOCK8(38, 42, 46, 50, 54, 58, 62, 34, 68) ::: CLOB);
        else if (VAR == 2)  // w class 0, q class 1
            asm volatile(BLOCK8(32, 36, 40, 44, 48, 52, 56, 60, 70) BLOCK8(36, 40, 44, 48, 52, 56, 60, 64, 70) ::: CLOB);
        else if (VAR == 3)  // w class 1, q class 1
            asm volatile(BLOCK8(34, 38, 42, 46, 50, 54, 58, 62, 70) BLOCK8(38, 42, 46, 50, 54, 58, 62, 34, 70) ::: CLOB);
        else                // w alternating classes as a sliding window does (w[a + t]), q class 0
            asm volatile(BLOCK8(32, 34, 36, 38, 40, 42, 44, 46, 68) BLOCK8(34, 36, 38, 40, 42, 44, 46, 48, 68) ::: CLOB);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) stamps[0] = c1 - c0, stamps[1] = r1 - r0;
    double s;
    asm volatile("v_add_f64 %0, v[0:1], v[4:5]\n" : "=v"(s)::CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VAR>
static void run(const char *name, int blocks_per_cu, int num_cu) {
    const int blocks = blocks_per_cu * num_cu, iters = 1000000;  // ~90 ms per launch: the clocks need time to settle
    double *out;
    unsigned long long *stamps, hs[2];
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipMalloc(&stamps, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(256), 0, 0, out, iters / 2, stamps);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(256), 0, 0, out, iters, stamps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 256 * iters * 16 * 2;
    hipMemcpy(hs, stamps, 16, hipMemcpyDeviceToHost);
    // s_memtime ticks = shader cycles, s_memrealtime = 100 MHz (MI355X_MICROARCH.md): cycles per v_fmac_f64 of ONE wave, clock
    printf("%-44s %d wave(s)/SIMD: %7.3f ms  %6.2f TFLOP/s  %5.2f cycles per FMA of a wave = %5.2f per SIMD, %4.2f GHz\n", name,
           blocks_per_cu, ms, flops / ms / 1e9, (double)hs[0] / ((double)iters * 16), (double)hs[0] / ((double)iters * 16) / blocks_per_cu,
           (double)hs[0] / (double)hs[1] * 0.1);
    hipFree(stamps);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cu = p.multiProcessorCount;
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("acc, w, q in the same register class", w, cu);
        run<1>("w in the other class", w, cu);
        run<2>("q in the other class", w, cu);
        run<3>("w and q in the other class", w, cu);
        run<4>("w alternating (sliding window), q same", w, cu);
    }
    return 0;
}
