// Microbenchmark (round 3): what limits the hot loop of kvm_ed_dot_kernel?  The correlation loop  acc[a] += X[c0 + a + j] * q[j]
// of one wavefront (8 rows x 8 lanes, CI candidates per lane, window chunk in wave-private LDS) in variants:
//   CI    : 8 (class A of the library) or 16 (128-candidate rows: half the LDS reads per FMA)
//   QMODE : 0 = query values by broadcast ds_read_b128 (the library), 1 = no query reads at all (two fixed registers: the bound
//           of any scheme that takes the query off the LDS), 2 = query values by scalar loads from the constant address space
//           (s_load -> SGPR operand of v_fmac_f64)
//   XMODE : 0 = window values by ds_read_b128 (the library), 1 = no window reads (registers only)
//   STAGE : 1 = a chunk's staging (27 / 35 global loads, row-shift subtraction, ds_write_b64) as in the library, 0 = none
// Occupancy is forced by the dynamic LDS size (WPS waves per SIMD = workgroups per CU), the grid is 16 rounds of the chip, so
// the dispatcher's placement cannot leave CUs idle.  Prints useful TFLOP/s (2 flop per FMA).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ed_loop_variants ed_loop_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
#define ROWS 8

// acc += q16[lane of my 16-lane row = K] * w : the query value rides in on the DPP row broadcast of the DP ALU (gfx90a+:
// v_fmac_f64_dpp supports row_newbcast only), so a VGPR pair that holds q[jb + (lane & 15)] serves 16 points with no LDS read.
#define BC_CASE(K) case K: asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(q16), "v"(w)); break;
__device__ __forceinline__ void fmac_bc(double &acc, double q16, double w, int k) {  // k: a constant after unrolling
    switch (k) {
        BC_CASE(0) BC_CASE(1) BC_CASE(2) BC_CASE(3) BC_CASE(4) BC_CASE(5) BC_CASE(6) BC_CASE(7)
        BC_CASE(8) BC_CASE(9) BC_CASE(10) BC_CASE(11) BC_CASE(12) BC_CASE(13) BC_CASE(14) BC_CASE(15)
    }
}

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int CI, int JC, int STAGE>
struct cfg {
    static constexpr int ROWLEN = CI * 8 + JC;          // staged points per row per chunk
    static constexpr int LOADS = (ROWLEN + 63) / 64;
    // register staging: rows a multiple of 256 B apart, displaced by 16 B x (r & 3) against each other; LDS-DMA staging (STAGE >= 2):
    // rows of whole 1-KiB pieces (128 doubles), 2064 B apart, which displaces consecutive rows by one 16-B slot
    static constexpr int PIECES = (ROWLEN + 127) / 128;
    static constexpr int RS = STAGE >= 2 ? PIECES * 128 + 2 : LOADS * 64 + 32;
    static constexpr int QLEN = STAGE >= 2 ? ((JC + 8 + 127) / 128) * 128 : 192;
    static constexpr int WAVE_LDS = (ROWS * RS + QLEN + 31) / 32 * 32;
    __device__ static constexpr int row_shift(int r) { return STAGE >= 2 ? 0 : (r & 3) * 2; }
};

template <int CI, int QMODE, int XMODE, int STAGE, int JC>
__global__ __launch_bounds__(256) void loop_kernel(const double *__restrict__ series, const double *__restrict__ qc_g,
                                                   double *__restrict__ out, int chunks, int n) {
    using C = cfg<CI, JC, STAGE>;
    extern __shared__ __attribute__((aligned(256))) d2 lds_all2[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane >> 3, lc = lane & 7;
    d2 *lds_wave2 = lds_all2 + wave * (C::WAVE_LDS / 2);
    double *wlds = reinterpret_cast<double *>(lds_wave2);
    double *qlds = wlds + ROWS * C::RS;
    const __attribute__((address_space(4))) double *qs = (const __attribute__((address_space(4))) double *)qc_g;
    // each row stages from its own position
    int64_t rstart[ROWS];
    double rshift[ROWS];
    const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 4099 % (n - 70000);
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        rstart[r] = base + r * 997;
        rshift[r] = series[rstart[r]];
    }
    double stage[ROWS][C::LOADS], qstage[3];
    auto load_chunk = [&](int j0) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const double *src = series + rstart[r] + j0;
#pragma unroll
            for (int t = 0; t < C::LOADS; t++) stage[r][t] = src[t * 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < 3; t++) qstage[t] = qc_g[j0 + t * 64 + lane];
    };
    auto store_chunk = [&]() {
        double sh[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) sh[r] = rshift[r];
        asm volatile("" : "+v"(sh[0]), "+v"(sh[1]), "+v"(sh[2]), "+v"(sh[3]), "+v"(sh[4]), "+v"(sh[5]), "+v"(sh[6]), "+v"(sh[7])::"memory");
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
#pragma unroll
            for (int t = 0; t < C::LOADS; t++) wlds[r * C::RS + C::row_shift(r) + t * 64 + lane] = stage[r][t] - sh[r];
        }
#pragma unroll
        for (int t = 0; t < 3; t++) qlds[t * 64 + lane] = qstage[t];
    };
    // LDS-DMA staging: piece p of every row (and of the query chunk) of the chunk that starts at j0, straight into the LDS
    auto dma_piece = [&](int j0, int p) {
#pragma unroll
        for (int r = 0; r < ROWS; r++)
            __builtin_amdgcn_global_load_lds((gptr_t *)(series + rstart[r] + j0 + p * 128 + 2 * lane), (lptr_t *)(wlds + r * C::RS + p * 128), 16, 0, 0);
        if (QMODE != 2 && p * 128 < C::QLEN)
            __builtin_amdgcn_global_load_lds((gptr_t *)(qc_g + j0 + p * 128 + 2 * lane), (lptr_t *)(qlds + p * 128), 16, 0, 0);
    };
    // fill the LDS once (also when STAGE == 0)
    if constexpr (STAGE >= 2) {
#pragma unroll
        for (int p = 0; p < C::PIECES; p++) dma_piece(0, p);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        load_chunk(0);
        store_chunk();
    }

    double acc[CI];
#pragma unroll
    for (int a = 0; a < CI; a++) acc[a] = 0.0;
    const d2 *X2 = reinterpret_cast<const d2 *>(lds_wave2) + (row * (C::RS / 2) + C::row_shift(row) / 2 + (CI / 2) * lc);
    const d2 *Q2 = reinterpret_cast<const d2 *>(lds_wave2) + ROWS * (C::RS / 2);
    constexpr int NP = CI / 2 + 2, RING = CI == 8 ? 8 : 16, U = RING / 2;
    static_assert(NP + 2 <= RING && (JC / 4) % U == 0, "ring");
    d2 fixq0 = {1.0 + 1e-3 * lane, -0.5}, fixq1 = {0.25, -1.25};
    d2 fixw[RING];
#pragma unroll
    for (int i = 0; i < RING; i++) fixw[i] = d2{1.0 + i * 1e-2, -1.0 + lane * 1e-3};
    for (int ch = 0; ch < chunks; ch++) {
        const int j0 = (ch * JC) & 8191;
        if (STAGE == 1) load_chunk(j0 + JC);
        d2 R[RING], Q[2][2];
        if (XMODE == 0) {
#pragma unroll
            for (int j = 0; j < NP; j++) R[j] = X2[j];
        } else {
#pragma unroll
            for (int j = 0; j < RING; j++) R[j] = fixw[j];
        }
        if (QMODE == 0) {
            Q[0][0] = Q2[0];
            Q[0][1] = Q2[1];
        }
        const double *q16p = qlds + (lane & 15);
        double qr[2], qrn[2];
        if (QMODE == 3) {
            qr[0] = q16p[0];
            qr[1] = q16p[16];
        }
        // scalar path: batches of 16 query values (two s_load_dwordx16), fetched one batch (4 sub-steps) ahead
        double qv[16], qn[16];
        if (QMODE == 2) {
#pragma unroll
            for (int k = 0; k < 16; k++) qv[k] = qs[j0 + k];
        }
        for (int body = 0; body < JC / 4 / U; body++) {
            if (STAGE == 3 && body * U == 32 + 4) dma_piece(j0 + JC, 0);   // (one body of margin: the ring has read ahead)
#pragma unroll
            for (int g = 0; g < U; g++) {
                const int G = body * U + g;
                if (QMODE == 3 && g == 0) {
                    qrn[0] = q16p[(body + 1) * 4 * U];
                    if (U == 8) qrn[1] = q16p[(body + 1) * 4 * U + 16];
                }
                if (QMODE == 2 && (g & 3) == 0) {
#pragma unroll
                    for (int k = 0; k < 16; k++) qn[k] = qs[j0 + (G / 4 + 1) * 16 + k];
                }
                if (XMODE == 0) {
                    R[(2 * g + NP) % RING] = X2[2 * G + NP];
                    R[(2 * g + NP + 1) % RING] = X2[2 * G + NP + 1];
                }
                if (QMODE == 0) {
                    Q[(g + 1) % 2][0] = Q2[2 * (G + 1)];
                    Q[(g + 1) % 2][1] = Q2[2 * (G + 1) + 1];
                }
                if constexpr (CI == 8)
                    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])::"memory");
                else
                    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                                 "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15])::"memory");
                double q0x, q0y, q1x, q1y;
                if (QMODE == 0) {
                    q0x = Q[g % 2][0].x, q0y = Q[g % 2][0].y, q1x = Q[g % 2][1].x, q1y = Q[g % 2][1].y;
                } else if (QMODE == 1) {
                    q0x = fixq0.x, q0y = fixq0.y, q1x = fixq1.x, q1y = fixq1.y;
                } else {
                    q0x = qv[4 * (g & 3)], q0y = qv[4 * (g & 3) + 1], q1x = qv[4 * (g & 3) + 2], q1y = qv[4 * (g & 3) + 3];
                }
#define W(k) (((k) & 1) ? R[(2 * g + (k) / 2) % RING].y : R[(2 * g + (k) / 2) % RING].x)
                if constexpr (QMODE == 3) {
                    const double qq = qr[g / 4];
                    const int K = 4 * (g & 3);
#pragma unroll
                    for (int a = 0; a < CI; a++) fmac_bc(acc[a], qq, W(a), K);
#pragma unroll
                    for (int a = 0; a < CI; a++) fmac_bc(acc[a], qq, W(a + 1), K + 1);
#pragma unroll
                    for (int a = 0; a < CI; a++) fmac_bc(acc[a], qq, W(a + 2), K + 2);
#pragma unroll
                    for (int a = 0; a < CI; a++) fmac_bc(acc[a], qq, W(a + 3), K + 3);
                } else {
#pragma unroll
                for (int a = 0; a < CI; a++) acc[a] = __builtin_fma(W(a), q0x, acc[a]);
#pragma unroll
                for (int a = 0; a < CI; a++) acc[a] = __builtin_fma(W(a + 1), q0y, acc[a]);
#pragma unroll
                for (int a = 0; a < CI; a++) acc[a] = __builtin_fma(W(a + 2), q1x, acc[a]);
#pragma unroll
                for (int a = 0; a < CI; a++) acc[a] = __builtin_fma(W(a + 3), q1y, acc[a]);
                }
#undef W
                if (QMODE == 3 && g == U - 1) {
                    qr[0] = qrn[0];
                    qr[1] = qrn[1];
                }
                if (QMODE == 2 && (g & 3) == 3) {
#pragma unroll
                    for (int k = 0; k < 16; k++) qv[k] = qn[k];
                }
            }
        }
        if (STAGE == 1) store_chunk();
        if constexpr (STAGE >= 2) {
            // the pieces that are still being read (all of them for STAGE 2; all but piece 0, issued early, for STAGE 3)
#pragma unroll
            for (int p = (STAGE == 3 ? 1 : 0); p < C::PIECES; p++) dma_piece(j0 + JC, p);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    double s = 0;
#pragma unroll
    for (int a = 0; a < CI; a++) s += acc[a];
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

static double *d_series, *d_q, *d_out;
static int num_cu;
static const int N = 1 << 24;

template <int CI, int QMODE, int XMODE, int STAGE, int JC = 128>
static void run(const char *name, int wps) {
    auto kern = loop_kernel<CI, QMODE, XMODE, STAGE, JC>;
    const size_t need = (size_t)4 * cfg<CI, JC, STAGE>::WAVE_LDS * 8;
    // occupancy: wps workgroups per CU by LDS size (160 KiB per CU)
    size_t lds = need;
    const size_t cap = 160 * 1024 / wps, floor_ = wps < 8 ? 160 * 1024 / (wps + 1) + 256 : 0;
    if (lds > cap) { printf("%-52s %d waves/SIMD: does not fit (%zu B of LDS per workgroup)\n", name, wps, need); return; }
    if (lds < floor_) lds = floor_;
    lds = (lds + 255) & ~(size_t)255;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = num_cu * wps * 8, chunks = 1024 * 128 / JC;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_series, d_q, d_out, chunks, N);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    const double flop = (double)blocks * 256 * chunks * JC * CI * 2;
    printf("%-52s %d waves/SIMD: %8.3f ms  %6.2f TFLOP/s%s\n", name, wps, best, flop / best / 1e9, err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    num_cu = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.gcnArchName, num_cu);
    std::vector<double> h(N + 4096), hq(8192 + 4096);
    srand(7);
    double x = 0;
    for (auto &v : h) { x += (rand() / (double)RAND_MAX - 0.5); v = x; }
    for (auto &v : hq) v = rand() / (double)RAND_MAX - 0.5;
    hipMalloc(&d_series, h.size() * 8); hipMalloc(&d_q, hq.size() * 8); hipMalloc(&d_out, (size_t)num_cu * 8 * 8 * 256 * 8);
    hipMemcpy(d_series, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_q, hq.data(), hq.size() * 8, hipMemcpyHostToDevice);
    for (int wps : {1, 2}) {
        run<8, 0, 0, 1>("CI 8  q LDS     x LDS  staged   (the library)", wps);
        run<8, 0, 0, 0>("CI 8  q LDS     x LDS  unstaged", wps);
        run<8, 1, 0, 0>("CI 8  q none    x LDS  unstaged", wps);
        run<8, 2, 0, 0>("CI 8  q scalar  x LDS  unstaged", wps);
        run<8, 2, 0, 1>("CI 8  q scalar  x LDS  staged", wps);
        run<8, 0, 0, 0, 192>("CI 8  q LDS     x LDS  unstaged        JC 192", wps);
        run<8, 0, 0, 1, 192>("CI 8  q LDS     x LDS  staged (regs)   JC 192", wps);
        run<8, 0, 0, 2, 192>("CI 8  q LDS     x LDS  LDS-DMA at end  JC 192", wps);
        run<8, 0, 0, 3, 192>("CI 8  q LDS     x LDS  LDS-DMA early   JC 192", wps);
        run<8, 0, 0, 2, 128>("CI 8  q LDS     x LDS  LDS-DMA at end  JC 128", wps);
        run<16, 0, 0, 2, 128>("CI 16 q LDS     x LDS  LDS-DMA at end  JC 128", wps);
        run<16, 2, 0, 2, 128>("CI 16 q scalar  x LDS  LDS-DMA at end  JC 128", wps);
        run<8, 3, 0, 0>("CI 8  q DPP     x LDS  unstaged", wps);
        run<8, 3, 0, 1>("CI 8  q DPP     x LDS  staged", wps);
        run<8, 3, 1, 0>("CI 8  q DPP     x none unstaged", wps);
        run<8, 0, 1, 0>("CI 8  q LDS     x none unstaged", wps);
        run<8, 1, 1, 0>("CI 8  q none    x none unstaged (bare, 8 acc)", wps);
        run<16, 0, 0, 1>("CI 16 q LDS     x LDS  staged", wps);
        run<16, 0, 0, 0>("CI 16 q LDS     x LDS  unstaged", wps);
        run<16, 1, 0, 0>("CI 16 q none    x LDS  unstaged", wps);
        run<16, 2, 0, 0>("CI 16 q scalar  x LDS  unstaged", wps);
        run<16, 2, 0, 1>("CI 16 q scalar  x LDS  staged", wps);
        run<16, 1, 1, 0>("CI 16 q none    x none unstaged (bare, 16 acc)", wps);
    }
    // occupancy: with chunks of 64 points a workgroup's LDS is 46 KB and three fit a CU (the library's 63 KB allow two)
    for (int wps : {1, 2, 3}) {
        run<8, 3, 0, 0, 64>("CI 8  q DPP     x LDS  unstaged  JC 64", wps);
        run<8, 3, 0, 1, 64>("CI 8  q DPP     x LDS  staged    JC 64", wps);
        run<8, 1, 0, 0, 64>("CI 8  q none    x LDS  unstaged  JC 64", wps);
        run<8, 1, 1, 0, 64>("CI 8  q none    x none unstaged  JC 64 (bare)", wps);
        run<4, 3, 0, 0, 64>("CI 4  q DPP     x LDS  unstaged  JC 64", wps);
    }
    return 0;
}
