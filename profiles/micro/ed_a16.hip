// Microbenchmark (round 5): would SIXTEEN candidates per lane pay for the full 64-candidate tiles of kvm_ed_dot_kernel?
// The library's class A is 8 rows x 8 lanes x 8 candidates: per FMA one byte comes back from the LDS into the VGPRs, and
// profiles/r03_ed_loop_variants.log showed that those returns (not the LDS array) take issue slots from the FMAs.  Here a
// wavefront is 16 rows x 4 lanes x 16 candidates -- the same 64-candidate tiles, half the bytes per FMA:
//   * row r, lane-in-row lc:  lc = lane >> 4,  r = (lane & 15) ^ ((lane >> 1) & 8), rows 1040 B apart: every hardware lane
//     group of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) then covers 16 distinct 16-B slots;
//   * a row of the wave's LDS is a plain circular ring of 128 points (no lookahead copy): a lane's read pointer is a VGPR
//     + immediate, and each of the four lane classes steps its pointer back by 1024 B at ONE known sub-step per revolution;
//   * stream staging as in the library: per 16-point block a lane fetches 32 B (4 points of its row) two blocks ahead of
//     the LDS write, which goes where the ring has gone dead;
//   * query values: QMODE 0 = one 8-byte global load per lane and block + v_fmac_f64_dpp row_newbcast (the library's
//     form, without the query ring in the LDS), 1 = scalar loads (s_load_dwordx16 -> SGPR operand), 2 = none (two fixed
//     registers: the bound).
// STAGE 0 drops the staging (the ring is filled once).  Occupancy: 2 workgroups of 4 waves per CU (66 560 B of LDS each).
// Prints useful TFLOP/s (2 flop per FMA); compare with ed_loop_variants' "CI 8 q DPP x LDS" lines and the library's
// full-tile rate on the same box.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ed_a16 ed_a16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

#include <utility>
template <int K>
__device__ __forceinline__ void fmac_bc(double &acc, double q16, double w) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(q16), "v"(w), "n"(K));
}
// compile-time loops: every register index and LDS offset below is a constant expression, not a constant "after unrolling"
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

#define CI 16
#define ROWS 16
#define ROW_B 1040                       // bytes between rows: 1024 of ring + one 16-B slot of skew
#define WAVE_B (ROWS * ROW_B)            // 16 640 B per wave
#define RINGP 16                         // register ring, pairs

template <int QMODE, int STAGE>
__global__ __launch_bounds__(256) void a16_kernel(const double *__restrict__ series, const double *__restrict__ qc_g,
                                                  double *__restrict__ out, int revs, int n) {
    extern __shared__ __attribute__((aligned(256))) char lds_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lc = lane >> 4, r = (lane & 15) ^ ((lane >> 1) & 8);
    char *wl = lds_all + wave * WAVE_B;
    char *rowp = wl + r * ROW_B;
    const __attribute__((address_space(4))) double *qs = (const __attribute__((address_space(4))) double *)qc_g;
    const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 4099 % (n - 70000);
    const int64_t rstart = base + r * 997;                  // each row stages from its own position
    const double c_row = series[rstart];
    const double *gsrc = series + rstart + 4 * lc;          // + 16 b + 128 rev: this lane's 4 points of block b
    char *wdst = rowp + 32 * lc;                            // + 128 b
    auto ld4 = [&](int pos, d2 &u, d2 &v) {
        const d2u a = *reinterpret_cast<const d2u *>(gsrc + pos), b = *reinterpret_cast<const d2u *>(gsrc + pos + 2);
        u = d2{a.x, a.y};
        v = d2{b.x, b.y};
    };
    auto wr4 = [&](int b, d2 u, d2 v) {
        u.x -= c_row; u.y -= c_row; v.x -= c_row; v.y -= c_row;
        *reinterpret_cast<d2 *>(wdst + 128 * b) = u;
        *reinterpret_cast<d2 *>(wdst + 128 * b + 16) = v;
    };
    // fill the ring with [0, 128); fetch blocks 8 and 9 (the next generation's blocks 0 and 1)
    d2 pend[2][2];
    {
        d2 u[8], v[8];
#pragma unroll
        for (int b = 0; b < 8; b++) ld4(16 * b, u[b], v[b]);
#pragma unroll
        for (int b = 0; b < 8; b++) wr4(b, u[b], v[b]);
        ld4(128, pend[0][0], pend[0][1]);
        ld4(144, pend[1][0], pend[1][1]);
    }
    double acc[CI];
#pragma unroll
    for (int a = 0; a < CI; a++) acc[a] = 0.0;
    // read pointer: pair k of a revolution is at P + 16 k; at the start P = row + 8 (16 lc + 20)
    const char *P0 = rowp + 128 * lc + 160;
    const char *P = P0;
    d2 R[RINGP];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 10; i++) R[i] = *reinterpret_cast<const d2 *>(rowp + 128 * lc + 16 * i);
    d2 fixq0 = {1.0 + 1e-3 * lane, -0.5}, fixq1 = {0.25, -1.25};
    double qq[2] = {0, 0};
    if (QMODE == 0) qq[0] = qc_g[lane & 15];
    double qv[16], qn[16];
    if (QMODE == 1) {
#pragma unroll
        for (int k = 0; k < 16; k++) qv[k] = qs[k];
    }
    for (int rev = 0; rev < revs; rev++) {
        const int j0 = (rev * 128) & 8191;
        static_for<32>([&](auto Gc) {
            constexpr int G = decltype(Gc)::value;
            if (G % 4 == 0) {
                const int b = G / 4;
                if (STAGE) {
                    // block b of the next generation goes where this generation's block b died two sub-steps ago; then
                    // fetch the block two further on
                    wr4(b, pend[b & 1][0], pend[b & 1][1]);
                    ld4(j0 + 128 + 16 * (b + 2), pend[b & 1][0], pend[b & 1][1]);
                }
                if (QMODE == 0) qq[(b + 1) & 1] = qc_g[j0 + 16 * (b + 1) + (lane & 15)];
                if (QMODE == 1) {
#pragma unroll
                    for (int k = 0; k < 16; k++) qn[k] = qs[j0 + 16 * (b + 1) + k];
                }
            }
            // the lane class whose pointer passes the end of the ring at this sub-step steps back by the ring's length
            if (G == 27 || G == 23 || G == 19 || G == 15) {
                const int cls = (27 - G) / 4;
                P = (lc == cls) ? P - 1024 : P;
            }
            R[(2 * G + 10) % RINGP] = *reinterpret_cast<const d2 *>(P + 16 * (2 * G));
            R[(2 * G + 11) % RINGP] = *reinterpret_cast<const d2 *>(P + 16 * (2 * G + 1));
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                         "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15])::"memory");
#define W(i) (((i) & 1) ? R[(2 * G + (i) / 2) % RINGP].y : R[(2 * G + (i) / 2) % RINGP].x)
            if constexpr (QMODE == 0) {
                const double q16 = qq[(G / 4) & 1];
                static_for<4>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    static_for<CI>([&](auto ac) {
                        constexpr int a = decltype(ac)::value;
                        fmac_bc<4 * (G & 3) + k>(acc[a], q16, W(a + k));
                    });
                });
            } else {
                double q4[4];
                if (QMODE == 1) {
                    q4[0] = qv[4 * (G & 3)], q4[1] = qv[4 * (G & 3) + 1], q4[2] = qv[4 * (G & 3) + 2], q4[3] = qv[4 * (G & 3) + 3];
                } else {
                    q4[0] = fixq0.x, q4[1] = fixq0.y, q4[2] = fixq1.x, q4[3] = fixq1.y;
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int a = 0; a < CI; a++) acc[a] = __builtin_fma(W(a + k), q4[k], acc[a]);
            }
#undef W
            if (QMODE == 1 && (G & 3) == 3) {
#pragma unroll
                for (int k = 0; k < 16; k++) qv[k] = qn[k];
            }
        });
        P = P0;   // (every lane has stepped back once: one revolution on, the pointers are where they started)
    }
    double s = 0;
#pragma unroll
    for (int a = 0; a < CI; a++) s += acc[a];
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- the same tiles with the Karatsuba split of the correlation (2-parallel fast FIR): for a pair of candidates (2u, 2u+1)
//     y[2u]   = EE(u) + OO(u),      y[2u+1] = SS(u) - OO(u) - EE(u+1)
//     EE(u) = sum_t x[2(u+t)] q[2t],  OO(u) = sum_t x[2(u+t)+1] q[2t+1],  SS(u) = sum_t (x[2(u+t)+1] + x[2(u+t)+2]) (q[2t] + q[2t+1])
// -- 25 accumulators per lane (9 EE, 8 OO, 8 SS) and 25 FMAs per TWO window points instead of 32, plus one add per two points
// for the sums of neighbouring window values (formed in registers: the LDS holds the plain window as before: a ds_read_b128
// returns (x[2n], x[2n+1]) = the even and the odd stream's entry n).  Rate printed as if 16 FMAs per point had been executed.
template <int STAGE>
__global__ __launch_bounds__(256, 2) void a16k_kernel(const double *__restrict__ series, const double *__restrict__ qe_g, const double *__restrict__ qo_g,
                                                   const double *__restrict__ qs_g, double *__restrict__ out, int revs, int n) {
    extern __shared__ __attribute__((aligned(256))) char lds_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lc = lane >> 4, r = (lane & 15) ^ ((lane >> 1) & 8);
    char *wl = lds_all + wave * WAVE_B;
    char *rowp = wl + r * ROW_B;
    const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 4099 % (n - 70000);
    const int64_t rstart = base + r * 997;
    const double c_row = series[rstart];
    const double *gsrc = series + rstart + 4 * lc;
    char *wdst = rowp + 32 * lc;
    auto ld4 = [&](int pos, d2 &u, d2 &v) {
        const d2u a = *reinterpret_cast<const d2u *>(gsrc + pos), b = *reinterpret_cast<const d2u *>(gsrc + pos + 2);
        u = d2{a.x, a.y};
        v = d2{b.x, b.y};
    };
    auto wr4 = [&](int b, d2 u, d2 v) {
        u.x -= c_row; u.y -= c_row; v.x -= c_row; v.y -= c_row;
        *reinterpret_cast<d2 *>(wdst + 128 * b) = u;
        *reinterpret_cast<d2 *>(wdst + 128 * b + 16) = v;
    };
    constexpr int AH = 2;
    d2 pend[AH][2];
    {
        d2 u[8], v[8];
#pragma unroll
        for (int b = 0; b < 8; b++) ld4(16 * b, u[b], v[b]);
#pragma unroll
        for (int i = 0; i < AH; i++) ld4(128 + 16 * i, pend[i][0], pend[i][1]);
#pragma unroll
        for (int b = 0; b < 8; b++) wr4(b, u[b], v[b]);
    }
    double EE[9], OO[8], SS[8];
#pragma unroll
    for (int a = 0; a < 9; a++) EE[a] = 0.0;
#pragma unroll
    for (int a = 0; a < 8; a++) OO[a] = SS[a] = 0.0;
    // pair p of the lane (window points 16 lc + 2p, + 1) sits at rowp + 128 lc + 16 p; sub-step S reads pairs 4S + 12 .. 4S + 15
    const char *P0 = rowp + 128 * lc + 192;
    const char *P = P0;
    d2 R[16];
    double XS[16];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 12; i++) R[i] = *reinterpret_cast<const d2 *>(rowp + 128 * lc + 16 * i);
#pragma unroll
    for (int i = 0; i < 11; i++) XS[i] = R[i].y + R[i + 1].x;
    double qe[2], qo[2], qs[2];
    qe[0] = qe_g[lane & 15]; qo[0] = qo_g[lane & 15]; qs[0] = qs_g[lane & 15];
    for (int rev = 0; rev < revs; rev++) {
        const int j0 = (rev * 128) & 8191, t0 = j0 / 2;
        static_for<16>([&](auto Sc) {
            constexpr int S = decltype(Sc)::value;
            if constexpr (S % 4 == 0) {   // the query values of the next 16 taps (32 points)
                constexpr int B = S / 4;
                qe[(B + 1) & 1] = qe_g[t0 + 16 * (B + 1) + (lane & 15)];
                qo[(B + 1) & 1] = qo_g[t0 + 16 * (B + 1) + (lane & 15)];
                qs[(B + 1) & 1] = qs_g[t0 + 16 * (B + 1) + (lane & 15)];
            }
            if constexpr (S % 2 == 0 && STAGE) {
                constexpr int b = S / 2;
                wr4(b, pend[b % AH][0], pend[b % AH][1]);
                ld4(j0 + 128 + 16 * (b + AH), pend[b % AH][0], pend[b % AH][1]);
            }
            if constexpr (S == 13 || S == 11 || S == 9 || S == 7) {
                constexpr int cls = (13 - S) / 2;
                P = (lc == cls) ? P - 1024 : P;
            }
            static_for<4>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                R[(4 * S + 12 + i) % 16] = *reinterpret_cast<const d2 *>(P + 64 * S + 16 * i);
            });
            asm volatile("" : "+v"(EE[0]), "+v"(EE[1]), "+v"(EE[2]), "+v"(EE[3]), "+v"(EE[4]), "+v"(EE[5]), "+v"(EE[6]), "+v"(EE[7]), "+v"(EE[8]),
                         "+v"(OO[0]), "+v"(OO[1]), "+v"(OO[2]), "+v"(OO[3]), "+v"(OO[4]), "+v"(OO[5]), "+v"(OO[6]), "+v"(OO[7])::"memory");
            const double qev = qe[(S / 4) & 1], qov = qo[(S / 4) & 1], qsv = qs[(S / 4) & 1];
            static_for<4>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int K = 4 * (S & 3) + k, T = 4 * S + k;   // DPP lane of the tap, tap index within the revolution
                static_for<9>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    fmac_bc<K>(EE[u], qev, R[(T + u) % 16].x);
                });
                static_for<8>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    fmac_bc<K>(OO[u], qov, R[(T + u) % 16].y);
                });
                static_for<8>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    fmac_bc<K>(SS[u], qsv, XS[(T + u) % 16]);
                });
            });
            // the sums the next sub-step adds: pairs 4S + 11 .. 4S + 15 have landed by now
            static_for<4>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                XS[(4 * S + 11 + i) % 16] = R[(4 * S + 11 + i) % 16].y + R[(4 * S + 12 + i) % 16].x;
            });
        });
        P = P0;
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) s += (EE[u] + OO[u]) + (SS[u] - OO[u] - EE[u + 1]);
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

static double *d_series, *d_q, *d_out;
static int num_cu;
static const int N = 1 << 24;

template <int QMODE, int STAGE>
static void run(const char *name, int wps) {
    auto kern = a16_kernel<QMODE, STAGE>;
    size_t lds = (size_t)4 * WAVE_B;
    const size_t cap = 160 * 1024 / wps, floor_ = wps < 8 ? 160 * 1024 / (wps + 1) + 256 : 0;
    if (lds > cap) { printf("%-52s %d waves/SIMD: does not fit\n", name, wps); return; }
    if (lds < floor_) lds = floor_;
    lds = (lds + 255) & ~(size_t)255;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = num_cu * wps * 8, revs = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_series, d_q, d_out, revs, N);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    const double flop = (double)blocks * 256 * revs * 128 * CI * 2;
    printf("%-52s %d waves/SIMD: %8.3f ms  %6.2f TFLOP/s%s\n", name, wps, best, flop / best / 1e9, err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}

template <int STAGE>
static void runk(const char *name, int wps) {
    auto kern = a16k_kernel<STAGE>;
    size_t lds = (size_t)4 * WAVE_B;
    const size_t cap = 160 * 1024 / wps, floor_ = wps < 8 ? 160 * 1024 / (wps + 1) + 256 : 0;
    if (lds > cap) { printf("%-52s %d waves/SIMD: does not fit\n", name, wps); return; }
    if (lds < floor_) lds = floor_;
    lds = (lds + 255) & ~(size_t)255;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = num_cu * wps * 8, revs = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_series, d_q, d_q + 4200, d_q + 8400, d_out, revs, N);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    const double flop = (double)blocks * 256 * revs * 128 * CI * 2;
    printf("%-52s %d waves/SIMD: %8.3f ms  %6.2f TFLOP/s (as if 16 FMAs per point)%s\n", name, wps, best, flop / best / 1e9, err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    num_cu = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.gcnArchName, num_cu);
    std::vector<double> h(N + 4096), hq(16384 + 4096);
    srand(7);
    double x = 0;
    for (auto &v : h) { x += (rand() / (double)RAND_MAX - 0.5); v = x; }
    for (auto &v : hq) v = rand() / (double)RAND_MAX - 0.5;
    hipMalloc(&d_series, h.size() * 8); hipMalloc(&d_q, hq.size() * 8); hipMalloc(&d_out, (size_t)num_cu * 8 * 8 * 256 * 8);
    hipMemcpy(d_series, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_q, hq.data(), hq.size() * 8, hipMemcpyHostToDevice);
    for (int wps : {1, 2}) {
        runk<1>("A16 Karatsuba q DPP(global) x ring staged", wps);
        runk<0>("A16 Karatsuba q DPP(global) x ring unstaged", wps);
        run<0, 1>("A16 q DPP(global) x ring staged", wps);
        run<0, 0>("A16 q DPP(global) x ring unstaged", wps);
        run<1, 1>("A16 q scalar      x ring staged", wps);
        run<1, 0>("A16 q scalar      x ring unstaged", wps);
        run<2, 1>("A16 q none        x ring staged", wps);
        run<2, 0>("A16 q none        x ring unstaged", wps);
    }
    return 0;
}
