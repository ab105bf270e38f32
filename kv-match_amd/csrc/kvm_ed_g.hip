// kvm_ed_g.hip -- the correlation-form ED pruner (see kvm_ed_dot.hip for the algebra and the guard band), lane-granular:
// RSM-ED with every point evaluated (KVM_OPT_EARLY_ABANDON = 0: the benchmark's headline mode), host-planned lists.
//
// What round 2's class A / B / C tiling loses on a phase-1-shaped list (runs of geometric length, mean 64): 16 % of the lane
// slots are empty (a run's remainder rounds up to 16 / 32 / 64 candidates) and the remainder classes cost 1.4-1.8 x a full
// tile per slot (4 accumulators per lane: LDS-bound) -- 15 % of the candidates take 28 % of the time -- and a chunk's
// staging (27 global loads, 27 subtractions of the row shift, 27 ds_write_b64 per 1024 FMAs) costs 18 % of the hot loop
// (profiles/micro/ed_loop_variants.hip).  Here:
//   * PACKING.  The list is cut into tiles of 8 consecutive candidates = one lane's accumulators (kvm_expand_tiles_kernel with
//     tile_max = 8, in run order).  A wavefront takes 64 consecutive tiles whatever runs they come from: lanes that follow
//     each other inside a run and inside a row of 8 lanes form a SEGMENT, which is staged once (8 k + chunk + 8 points for k
//     lanes) at an LDS address congruent to 4 i + (row & 3) slots of 16 B mod 16 (i = its first lane in the row): every 16-lane
//     group of ds_read_b128 then covers 16 distinct slots, exactly as the fixed rows of kvm_ed_dot.hip do.  Only the last
//     lane of a run is partly empty: 95 % of the slots are filled on the bench list (84 % before), all of them in the
//     8-accumulator loop.  The chunk length adapts to the number of segments of the unit (16 .. 128 points in steps of 16:
//     the wave's 20 KB of LDS hold 64 x 8 candidates + one chunk (+ pad) per segment).
//   * STAGING by LDS-DMA (global_load_lds_dwordx4): the unit's LDS image is lane-linear, 1 KiB per instruction, and every
//     lane's SOURCE address is its own -- the packing is nothing but a per-lane table of source offsets (<= 19 registers,
//     built once per unit).  No register, no VALU and no ds_write between HBM and LDS.  The row shift that centres the data
//     (x' = x - c) cannot be applied on the way, so the source is a copy of the series that is already centred per prefix
//     block (kvm_ctx::d_series_sh = x - first point of the block); a window that crosses into another block is re-based
//     afterwards: sum (x - c_row) qc = acc + sum_b (c_b - c_row) (qc_pre[end_b] - qc_pre[start_b]).
//   * BALANCE by construction ("stream-K"): the launch is exactly one wavefront per wave slot of the chip, and the work --
//     units x window groups of 16 points -- is cut into equally long contiguous ranges, one per wavefront.  A range that
//     crosses a unit boundary makes two partial passes; the partial accumulators of a unit meet through global memory
//     (sc1 stores, one agent-scope counter per unit, the wavefront whose add comes last finishes the unit: no wave ever
//     waits for another one).  No rounds, no tail: 2 063 units on 2 048 slots cost 1.007 units' time, not 2.
#include "kvm_internal.h"
#include "kvm_seqadd.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double edg_d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void edg_gptr;
typedef __attribute__((address_space(3))) void edg_lptr;

// EDG_DB = 1: ONE wavefront per SIMD with 40 KB of LDS and two images -- the LDS-DMA of chunk t+1 is issued before chunk t is
// computed and has the whole chunk to land (a chunk's image is re-fetched from L2 / HBM every 80-128 points: with one image
// the wait for it is a quarter of the chunk loop, in-kernel stamps); EDG_DB = 0: two wavefronts per SIMD, one image each.
#ifndef EDG_DB
#define EDG_DB 0
#endif
#define EDG_TMAX (EDG_DB ? 18 : 17)                         // LDS-DMA instructions (1 KiB = 128 doubles each) for the window regions
#define EDG_BUDGET (EDG_TMAX * 128)                         // doubles for the window regions: whole DMA pieces, so that the last
                                                            // instruction of an image cannot write into the query region
#define EDG_QLEN 160                                        // query region: a chunk (<= 128 points) + the loop's read-ahead (a
                                                            // 1-KiB piece + a 256-B piece)
#define EDG_NBUF (EDG_DB ? 2 : 1)
#define EDG_WAVE_LDS (EDG_NBUF * (EDG_BUDGET + EDG_QLEN) + 32)  // doubles per wavefront: 38.6 KB / 18.5 KB
#define EDG_WPS (EDG_DB ? 1 : 2)                            // wavefronts per SIMD = workgroups per CU
#define EDG_INLINE_MAX 4
static_assert(EDG_BUDGET % 32 == 0 && EDG_WAVE_LDS % 32 == 0 && 4 * EDG_WAVE_LDS * 8 * EDG_WPS <= 163840, "LDS layout");

struct edg_args {
    const double *series, *series_sh;
    const double2 *prefix;
    kvm_query_dev qd;
    kvm_surv *surv;
    int64_t cap;
    unsigned long long *counters;
    int32_t *out_off;
    double *out_dist;
    int64_t out_cap;
    // stream-K: partial accumulators [slot][a][lane] and one arrival counter per unit
    double *parts;
    unsigned *arrived;
    int64_t units;
    const int64_t *unit_off;  // [units + 1] exclusive prefix sums of the units' modelled costs (edg_unit_cost); [units] = the total
    // per unit, built once by kvm_ed_g_plan_kernel: the LDS-DMA source offsets [unit][t][lane], every lane's read position in the
    // LDS image (16-B slots), chunk length and number of DMA instructions
    const int32_t *tab_soff, *tab_x2;
    const int32_t *tab_pf;    // [unit][2][lane] series offsets (points, relative to a chunk's j0) of the 128-B lines the NEXT chunk's image
                              // brings in for the first time (-1: none): touched one chunk ahead, so that the LDS-DMA hits the L2
    const int2 *tab_meta;
    unsigned long long *phase_t;  // experiments (KVM_EDG_DBG & 2): summed 100-MHz ticks of [0] table loads [1] chunk loops [2] hand-over
                                  // + decision [3] queue pops + unit search [4] passes [5] whole wavefronts
    unsigned *next_piece;     // the work queue: pieces = equal ranges of the cost axis, popped in order
    int64_t n_pieces;
    int groups_per_unit, n_waves;
    int dbg;      // experiments (KVM_EDG_DBG; answers are wrong with any bit set): 1 = no LDS-DMA after a pass's first chunk
};

// LDS-DMA by inline asm: with __builtin_amdgcn_global_load_lds in the function hipcc (ROCm 7.2) waits `lgkmcnt(0)` in front of
// every use of a ds_read result -- also for the reads the hot loop has just issued for its NEXT sub-step, which exposes one LDS
// latency per 64 FMAs (seen in the ISA; the same loop without the builtin gets counted waits).  An asm load is absent from
// the compiler's bookkeeping: the waits for it (`s_waitcnt vmcnt(0)` before the first ds_read of a chunk) are written by hand.
// M0 = the wave-uniform LDS byte address of the 1-KiB (256-B) piece; lane l's 16 (4) bytes land at M0 + 16 l (4 l).
__device__ __forceinline__ void edg_dma16(const void *gsrc, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte)
                 : "memory");
}
__device__ __forceinline__ void edg_dma4(const void *gsrc, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte)
                 : "memory");
}

// Stream-K over COST, not over points: a unit with many segments has short chunks (the LDS holds one chunk per segment) and
// every chunk pays a fixed price -- the wait for its LDS-DMA image, the ring's start-up -- so its window costs up to 1.6 x a unit
// of long runs (in-kernel stamps: 3.5 us per chunk of 80 points, of which 2.4 are FMAs).  Cost of a unit, in points:
// m + EDG_CHUNK_COST per chunk.  The wavefronts cut the prefix sums of the costs into equal ranges; inside a unit cost and window
// position are proportional.
#define EDG_CHUNK_COST 36
__device__ __forceinline__ int edg_chunk_len(int nseg, int nlanes, bool &aligned) {
    // the largest multiple of 16 (<= 128) the unit's segments fit the LDS with -- with the bank-spreading alignment of the
    // regions (<= 15 slots of pad each) if that leaves >= 32 points, without it otherwise
    aligned = true;
    int JC = ((EDG_BUDGET - 8 * nlanes - nseg * (8 + 30)) / nseg) & ~15;
    if (JC < 32) {
        aligned = false;
        JC = ((EDG_BUDGET - 8 * nlanes - nseg * 8) / nseg) & ~15;
    }
    return JC > 128 ? 128 : JC;  // (>= 16: nseg <= 64, nlanes <= 64)
}
__device__ __forceinline__ int64_t edg_unit_cost(int m, int JC, int chunk_cost) { return (int64_t)m + (int64_t)((m + JC - 1) / JC) * chunk_cost; }
// the wavefront whose cost range [floor(w W / nw), floor((w + 1) W / nw)) holds cost position x
__device__ __forceinline__ int64_t edg_wave_of(int64_t x, int64_t total, int64_t nw) { return ((x + 1) * nw - 1) / total; }
// first window group (16 points) of the part of a unit that starts at cost position p (unit: [off, off + cost), gpu groups)
__device__ __forceinline__ int edg_group_at(int64_t p, int64_t off, int64_t cost, int gpu) {
    if (p <= off) return 0;
    if (p >= off + cost) return gpu;
    return (int)(((p - off) * gpu + cost - 1) / cost);
}
// segments of a unit: lanes that follow each other inside a run and inside a row of 8 lanes
__device__ __forceinline__ void edg_segments(const kvm_tile &mine, int lane, unsigned long long &heads, unsigned long long &valids) {
    const bool valid = mine.count > 0;
    const int p_start = __shfl_up(mine.start, 1, 64), p_run = __shfl_up(mine.run, 1, 64), p_valid = __shfl_up((int)valid, 1, 64);
    const bool cont = valid && (lane & 7) != 0 && p_valid && p_run == mine.run && p_start + 8 == mine.start;
    heads = __ballot(valid && !cont);
    valids = __ballot(valid);
}

// One partial pass: unit `unit` (tiles [64 unit, 64 unit + 64)), window points [16 r_lo, min(m, 16 r_hi)).
__device__ __forceinline__ void edg_pass(const edg_args &A, const kvm_tile *__restrict__ tiles, int64_t n_tiles, int64_t unit,
                                         int r_lo, int r_hi, double2 *lds_wave2, int64_t my_wave) {
    const int lane = threadIdx.x & 63;
    const unsigned long long ts0 = A.phase_t ? wall_clock64() : 0;
    const kvm_query_dev &qd = A.qd;
    const int m = qd.m;
    const double eps2 = qd.eps2;
    const double *__restrict__ series = A.series;
    const double *__restrict__ qc = qd.qc;
    double *wlds = reinterpret_cast<double *>(lds_wave2);
    const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(edg_lptr *)wlds);  // LDS byte address

    // ---- the unit's staging table (kvm_ed_g_plan_kernel) -------------------------------------------------------------------
    const int2 meta = A.tab_meta[unit];
    const int JC = __builtin_amdgcn_readfirstlane(meta.x), T = __builtin_amdgcn_readfirstlane(meta.y);
    if (T == 0) return;  // (no tile at all)
    int32_t soff[EDG_TMAX];
    {
        const int32_t *__restrict__ ts = A.tab_soff + ((size_t)unit * EDG_TMAX) * 64 + lane;
#pragma unroll
        for (int t = 0; t < EDG_TMAX; t++) soff[t] = ts[t * 64];   // (unconditional: one load latency for the lot)
    }
    const int x2_off = A.tab_x2[unit * 64 + lane];
    const int32_t pf0 = A.tab_pf[(unit * 2 + 0) * 64 + lane], pf1 = A.tab_pf[(unit * 2 + 1) * 64 + lane];

    // LDS of the wavefront: image 0 | image 1 | query chunk 0 | query chunk 1   (EDG_DB = 0: image | query chunk)
    const edg_d2 *X2b = reinterpret_cast<const edg_d2 *>(lds_wave2) + x2_off;
    const edg_d2 *Q2b = reinterpret_cast<const edg_d2 *>(lds_wave2) + (EDG_NBUF * EDG_BUDGET / 2);

    // ---- chunks of JC points from 16 r_lo to min(m, 16 r_hi) ----------------------------------------------------------------
    double acc[8];
#pragma unroll
    for (int a = 0; a < 8; a++) acc[a] = 0.0;
    const int j_end = min(m, 16 * r_hi);
    const unsigned long long ts1 = A.phase_t ? wall_clock64() : 0;
    // issue the LDS-DMA of the chunk that starts at j0 into image `buf`: the unit's image (T x 1 KiB) and the query chunk
    auto dma_chunk = [&](int j0, int buf) {
        const double *__restrict__ src = A.series_sh + j0;
        const unsigned img = lds_base + (unsigned)buf * (EDG_BUDGET * 8u), qb = lds_base + (EDG_NBUF * EDG_BUDGET + buf * EDG_QLEN) * 8u;
#pragma unroll
        for (int t = 0; t < EDG_TMAX; t++)
            if (t < T) edg_dma16(src + soff[t], img + 1024u * t);
        edg_dma16(qc + j0 + 2 * lane, qb);
        if (JC > 112)  // (the read-ahead beyond 128 points: 64 lanes x 4 B = 32 doubles)
            edg_dma4(reinterpret_cast<const float *>(qc + j0 + 128) + lane, qb + 128 * 8u);
    };
    const int n_dma = T + 1 + (JC > 112 ? 1 : 0);  // instructions per chunk (the counted wait below leaves the NEXT chunk's in flight)
    if (EDG_DB && 16 * r_lo < j_end) dma_chunk(16 * r_lo, 0);
    int buf = 0;
    unsigned pf_sink0 = 0, pf_sink1 = 0;
    for (int j0 = 16 * r_lo; j0 < j_end; j0 += JC, buf ^= (EDG_NBUF - 1)) {
        if (EDG_DB) {
            // the next chunk's image goes into the other buffer (last read a chunk ago) while this one is computed
            if (j0 + JC < j_end) {
                dma_chunk(j0 + JC, buf ^ 1);
                switch (n_dma) {   // s_waitcnt takes an immediate: this chunk's image has landed, the next one's may still be on its way
#define EDG_WAIT(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
                    EDG_WAIT(2) EDG_WAIT(3) EDG_WAIT(4) EDG_WAIT(5) EDG_WAIT(6) EDG_WAIT(7) EDG_WAIT(8) EDG_WAIT(9) EDG_WAIT(10) EDG_WAIT(11)
                    EDG_WAIT(12) EDG_WAIT(13) EDG_WAIT(14) EDG_WAIT(15) EDG_WAIT(16) EDG_WAIT(17) EDG_WAIT(18) EDG_WAIT(19) EDG_WAIT(20)
#undef EDG_WAIT
                    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            // one image: nothing of the previous chunk is still being read (its last FMAs have consumed every ds_read)
            if (!(A.dbg & 1) || j0 == 16 * r_lo) dma_chunk(j0, 0);
            if (!(A.dbg & 4)) {
                // two plain loads per lane pull the lines the next chunk's image adds into the L2 (results unused; issued AFTER
                // the DMA, so the counted wait below leaves them in flight)
                // (the two sink registers stay live -- "+v", loop-carried -- until the wait behind the loop: a load still in
                // flight must not find its destination handed to something else)
                const double *__restrict__ nx = A.series_sh + j0;
                asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink0) : "v"(nx + (pf0 < 0 ? 0 : pf0)) : "memory");
                asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink1) : "v"(nx + (pf1 < 0 ? 0 : pf1)) : "memory");
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        const edg_d2 *X2 = X2b + buf * (EDG_BUDGET / 2), *Q2 = Q2b + buf * (EDG_QLEN / 2);
        double *qlds = wlds + EDG_NBUF * EDG_BUDGET + buf * EDG_QLEN;
        const int jlim = min(JC, j_end - j0);
        const int bodies = jlim >> 4;
        if (bodies > 0) {
            // the register ring of kvm_ed_dot.hip (class A): 8 pairs, sub-step g reads pairs 2g .. 2g+5 and issues the reads of
            // sub-step g+1 before its own 32 FMAs
            constexpr int NP = 6, RING = 8, U = 4;
            edg_d2 R[RING], Q[2][2];
#pragma unroll
            for (int j = 0; j < NP; j++) R[j] = X2[j];
            Q[0][0] = Q2[0];
            Q[0][1] = Q2[1];
            for (int body = 0; body < bodies; body++) {
#pragma unroll
                for (int g = 0; g < U; g++) {
                    const int G = body * U + g;
                    R[(2 * g + NP) % RING] = X2[2 * G + NP];
                    R[(2 * g + NP + 1) % RING] = X2[2 * G + NP + 1];
                    Q[(g + 1) % 2][0] = Q2[2 * (G + 1)];
                    Q[(g + 1) % 2][1] = Q2[2 * (G + 1) + 1];
                    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),
                                 "+v"(acc[6]), "+v"(acc[7])::"memory");
                    const edg_d2 q0 = Q[g % 2][0], q1 = Q[g % 2][1];
#define EDG_W(k) (((k) & 1) ? R[(2 * g + (k) / 2) % RING].y : R[(2 * g + (k) / 2) % RING].x)
#pragma unroll
                    for (int a = 0; a < 8; a++) acc[a] = __builtin_fma(EDG_W(a), q0.x, acc[a]);
#pragma unroll
                    for (int a = 0; a < 8; a++) acc[a] = __builtin_fma(EDG_W(a + 1), q0.y, acc[a]);
#pragma unroll
                    for (int a = 0; a < 8; a++) acc[a] = __builtin_fma(EDG_W(a + 2), q1.x, acc[a]);
#pragma unroll
                    for (int a = 0; a < 8; a++) acc[a] = __builtin_fma(EDG_W(a + 3), q1.y, acc[a]);
#undef EDG_W
                }
            }
        }
        if (jlim & 15) {  // (the window's last points, m not a multiple of 16)
            const double *X = reinterpret_cast<const double *>(X2);
            for (int jj = bodies << 4; jj < jlim; jj++) {
                const double qv = qlds[jj];
#pragma unroll
                for (int a = 0; a < 8; a++) acc[a] = __builtin_fma(X[jj + a], qv, acc[a]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (read-ahead of the ring: retired before the next chunk's DMA)
    }

    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink0), "+v"(pf_sink1)::"memory");  // (the last prefetches have landed: sinks released)
    const unsigned long long ts2 = A.phase_t ? wall_clock64() : 0;
    struct edg_stamp {  // (records the pass's phases on every way out)
        const edg_args &A; unsigned long long t0, t1, t2; int lane;
        __device__ ~edg_stamp() {
            if (A.phase_t && lane == 0) {
                const unsigned long long t3 = wall_clock64();
                atomicAdd(A.phase_t + 0, t1 - t0); atomicAdd(A.phase_t + 1, t2 - t1); atomicAdd(A.phase_t + 2, t3 - t2); atomicAdd(A.phase_t + 4, 1ULL);
            }
        }
    } stamp{A, ts0, ts1, ts2, lane};
    // ---- the unit's passes meet: all but the last to arrive leave their accumulators in global memory ---------------------
    const int64_t u_off = A.unit_off[unit], u_end = A.unit_off[unit + 1], total_cost = A.unit_off[A.units];
    const int64_t w_first = edg_wave_of(u_off, total_cost, A.n_pieces), w_last = edg_wave_of(u_end - 1, total_cost, A.n_pieces);
    if (w_last > w_first) {
        double *mine_slot = A.parts + ((size_t)(unit + my_wave) * 8 * 64);
#pragma unroll
        for (int a = 0; a < 8; a++) __hip_atomic_store(mine_slot + a * 64 + lane, acc[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned seen = 0;
        if (lane == 0) seen = __hip_atomic_fetch_add(A.arrived + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
        if (seen != (unsigned)(w_last - w_first)) return;  // somebody else finishes this unit
        // the last to arrive: every pass in wave order (this one's from the same slot: one summation order whoever comes last)
#pragma unroll
        for (int a = 0; a < 8; a++) acc[a] = 0.0;
        for (int64_t w = w_first; w <= w_last; w++) {
            const double *slot = A.parts + ((size_t)(unit + w) * 8 * 64);
#pragma unroll
            for (int a = 0; a < 8; a++) acc[a] += __hip_atomic_load(slot + a * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- decide: correlation form of sum (x - q)^2 with its guard band (kvm_ed_dot.hip) ----------------------------------
    kvm_tile mine;
    mine.start = 1; mine.count = 0; mine.run_start = 1; mine.run = -1;
    if (unit * 64 + lane < n_tiles) mine = tiles[unit * 64 + lane];
    const double2 *__restrict__ prefix = A.prefix;
    const double qbar = qd.q_mean;
    const double2 qtot = qd.q_pref16[(m + 15) / 16];
    const int64_t row_block = ((int64_t)mine.start - 1) >> KVM_PS_BLOCK_LOG2;
    const double c_row = series[row_block << KVM_PS_BLOCK_LOG2];  // the shift the staged values of this lane's block carry
    unsigned keepm = 0;
#pragma unroll
    for (int a = 0; a < 8; a++) {
        if (a < mine.count) {
            const int64_t w0 = (int64_t)mine.start - 1 + a;
            double s1, s2, cb, pm;
            kvm_window_shifted_sums(prefix, series, w0, m, s1, s2, cb, &pm);
            const double dc = cb - c_row;
            const kvm_sum_err e = kvm_prefix_err(pm);
            const double S1 = s1 + m * dc, S2 = s2 + (2.0 * dc) * s1 + m * (dc * dc);
            const double E1 = e.e1 + 1e-15 * (fabs(s1) + m * fabs(dc));
            const double E2 = e.e2 + 2.0 * fabs(dc) * e.e1 + 1e-15 * (fabs(s2) + 2.0 * fabs(dc * s1) + m * (dc * dc));
            // points of the window that lie in another prefix block were staged with that block's shift
            double corr = 0.0, corr_mag = 0.0;
            for (int64_t b = w0 >> KVM_PS_BLOCK_LOG2; b <= (w0 + m - 1) >> KVM_PS_BLOCK_LOG2; b++) {
                if (b == row_block) continue;
                const int64_t lo = max((int64_t)0, (b << KVM_PS_BLOCK_LOG2) - w0), hi = min((int64_t)m, ((b + 1) << KVM_PS_BLOCK_LOG2) - w0);
                const double dcb = series[b << KVM_PS_BLOCK_LOG2] - c_row;
                corr += dcb * (qd.qc_pre[hi] - qd.qc_pre[lo]);
                corr_mag += fabs(dcb);
            }
            const double accp = acc[a] + corr;
            const double d = qbar - c_row;
            const double cross = accp + d * S1;
            const double qq = qtot.y + (2.0 * d) * qtot.x + m * (d * d);
            // (qc_pre: m running additions of |qc| <= sqrt(m sum qc^2) by Cauchy-Schwarz, 1.2e-16 each)
            const double slack = eps2 * 1e-9 + 1e-11 * (fabs(S2) + 2.0 * fabs(accp) + 2.0 * fabs(corr) + 2.0 * fabs(d * S1) + fabs(qq)) +
                                 E2 + 2.0 * fabs(d) * E1 + 1e-15 * m * sqrt(m * qtot.y) * corr_mag;
            const double dist = S2 - 2.0 * cross + qq;
            if (!(dist > eps2 + slack)) keepm |= 1u << a;  // (NaN: a poisoned prefix block -> keep)
        }
    }
    if (A.out_off) {
        // a few survivors: evaluated exactly by this wave, in place (K/QueryEngine.java:355-358: terms (x-q)*(x-q) in order)
        int ns = __popc(keepm);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ns += __shfl_xor(ns, o, 64);
        if (ns > 0 && ns <= EDG_INLINE_MAX) {
            const double *__restrict__ qx = qd.q;
            for (int it = 0; it < ns; it++) {
                const unsigned long long owners = __ballot(keepm != 0);
                const int L = __ffsll((long long)owners) - 1;
                const int a_me = keepm ? __ffs((int)keepm) - 1 : 0;
                const int start = __shfl(mine.start + a_me, L, 64);
                if (lane == L) keepm &= keepm - 1;
                const double *__restrict__ xs = series + ((int64_t)start - 1);
                double dist = 0.0;
                for (int jb = 0; jb < m; jb += 1024) {
                    const int cnt = min(1024, m - jb);
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int j = jb + i * 64 + lane;
                        const double xv = xs[j < m ? j : m - 1], qv = qx[j < m ? j : m - 1];
                        wlds[i * 64 + lane] = (xv - qv) * (xv - qv);               // K/QueryEngine.java:357
                    }
                    if (lane == 0) dist = ex_seq_add(wlds, cnt, dist);              // :357, strictly in order
                }
                if (lane == 0) {
                    atomicAdd(&A.counters[KVM_CNT_RECHECKED], 1ULL);
                    if (dist <= eps2) {
                        const unsigned long long o = atomicAdd(&A.counters[KVM_CNT_ANSWERS], 1ULL);
                        if ((int64_t)o < A.out_cap) {
                            A.out_off[o] = start;
                            A.out_dist[o] = sqrt(dist);
                        }
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int a = 0; a < 8; a++) {
        if ((keepm >> a) & 1u) {
            const unsigned long long idx = atomicAdd(&A.counters[KVM_CNT_CANDS1], 1ULL);
            if ((int64_t)idx < A.cap) A.surv[idx] = kvm_surv{mine.start + a, mine.run};
        }
    }
}

__global__ __launch_bounds__(256, EDG_WPS) void kvm_ed_g_kernel(edg_args A, const kvm_tile *__restrict__ tiles, int64_t n_tiles) {
    __shared__ __attribute__((aligned(256))) double2 lds_all2[4 * EDG_WAVE_LDS / 2];
    // (the wave index must LOOK uniform to the compiler: everything the chunk loop's control depends on derives from it, and a
    // "divergent" loop gets exec-mask control flow, loop counters in VGPRs and lgkmcnt(0) waits in the hot loop)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    double2 *lds_wave2 = lds_all2 + wave * (EDG_WAVE_LDS / 2);
    const int64_t total_cost = A.unit_off[A.units];
    const int gpu = A.groups_per_unit;
    // Pieces = equal ranges of the cost axis, several per wavefront, popped in order from one counter: two wavefronts with the
    // same modelled cost finish 25 % apart (in-kernel stamps: memory and the neighbour on the SIMD decide), so the cut is not
    // made in advance.  Every wavefront ends when the queue is empty; nobody waits for anybody.
    const unsigned long long tw0 = A.phase_t ? wall_clock64() : 0;
    unsigned long long tq = 0;
    for (int guard = 0; guard < (1 << 24); guard++) {
        const unsigned long long tq0 = A.phase_t ? wall_clock64() : 0;
        unsigned pc = 0;
        if (lane == 0) pc = atomicAdd(A.next_piece, 1u);
        const int64_t piece = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)pc);
        if (piece >= A.n_pieces) break;
        const int64_t p_lo = piece * total_cost / A.n_pieces, p_hi = (piece + 1) * total_cost / A.n_pieces;
        if (p_hi <= p_lo) continue;
        // first unit of the piece: the last one whose offset is <= p_lo -- 64 probes per step
        int64_t lo = 0, hi = A.units;  // invariant: unit_off[lo] <= p_lo < unit_off[hi]
        while (hi - lo > 1) {
            const int64_t span = hi - lo, stride = (span + 63) / 64;
            const int64_t probe = min(hi, lo + (int64_t)lane * stride);
            const unsigned long long le = __ballot(probe < hi && A.unit_off[probe] <= p_lo);   // (a prefix of the lanes: offsets ascend)
            const int last = 63 - __clzll((long long)le);                                      // (lane 0 probes `lo`: always set)
            const int64_t nlo = lo + (int64_t)last * stride;
            hi = min(hi, nlo + stride);
            lo = nlo;
        }
        if (A.phase_t) tq += wall_clock64() - tq0;
        for (int64_t unit = lo; unit < A.units; unit++) {
            const int64_t off = A.unit_off[unit], end = A.unit_off[unit + 1];
            if (off >= p_hi) break;
            const int r_lo = edg_group_at(p_lo, off, end - off, gpu), r_hi = edg_group_at(p_hi, off, end - off, gpu);
            edg_pass(A, tiles, n_tiles, unit, r_lo, r_hi, lds_wave2, piece);   // (an empty range still takes part in the hand-over)
        }
    }
    if (A.phase_t && lane == 0) {
        atomicAdd(A.phase_t + 3, tq);
        atomicAdd(A.phase_t + 5, wall_clock64() - tw0);
    }
}

// One wavefront per unit, once per query: the unit's segments, its chunk length, the layout of its LDS image -- as the table
// of LDS-DMA source offsets and every lane's read position -- and its cost.
__global__ __launch_bounds__(256) void kvm_ed_g_plan_kernel(const kvm_tile *__restrict__ tiles, int64_t n_tiles, int64_t units, int m,
                                                            int chunk_cost, int64_t *__restrict__ cost, int32_t *__restrict__ tab_soff,
                                                            int32_t *__restrict__ tab_x2, int2 *__restrict__ tab_meta,
                                                            int32_t *__restrict__ tab_pf) {
    const int lane = threadIdx.x & 63;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= units) return;
    kvm_tile mine;
    mine.start = 1; mine.count = 0; mine.run_start = 1; mine.run = -1;
    if (unit * 64 + lane < n_tiles) mine = tiles[unit * 64 + lane];
    const bool valid = mine.count > 0;
    unsigned long long heads, valids;
    edg_segments(mine, lane, heads, valids);
    const int nseg = __popcll(heads), nlanes = __popcll(valids);
    if (nseg == 0) {
        if (lane == 0) {
            cost[unit] = 1;
            tab_meta[unit] = make_int2(16, 0);
        }
        tab_pf[(unit * 2 + 0) * 64 + lane] = -1;
        tab_pf[(unit * 2 + 1) * 64 + lane] = -1;
        return;
    }
    const int my_head = valid ? 63 - __clzll((long long)(heads & ((2ULL << lane) - 1ULL))) : 0;
    bool aligned;
    const int JC = edg_chunk_len(nseg, nlanes, aligned);
    // regions in lane order (slots of 16 B); each lane notes its own segment's base and, for each of the DMA instructions,
    // the series offset (in points) of the slot it will fetch (0 for the pads between the regions: any valid address)
    int32_t soff[EDG_TMAX];
#pragma unroll
    for (int t = 0; t < EDG_TMAX; t++) soff[t] = 0;
    int cur = 0, my_base = 0;
    int32_t pf[2] = {-1, -1};
    int pf_next = 0;   // entries handed out so far (entry e belongs to lane e & 63, slot e >> 6)
    for (unsigned long long hb = heads; hb;) {
        const int h = __ffsll((long long)hb) - 1;
        hb &= hb - 1;
        const unsigned long long stop = (heads | ~valids) & ~((2ULL << h) - 1ULL);
        int nxt = stop ? __ffsll((long long)stop) - 1 : 64;
        nxt = min(nxt, (h | 7) + 1);
        const int k = nxt - h;
        const int gstart = __shfl(mine.start, h, 64) - 1;
        int base = cur;
        if (aligned) base += ((4 * (h & 7) + ((h >> 3) & 3)) - cur) & 15;
        const int size = (8 * k + JC + 8) >> 1;
        if (valid && my_head == h) my_base = base;
        const int t0 = base >> 6, t1 = (base + size - 1) >> 6;
#pragma unroll
        for (int t = 0; t < EDG_TMAX; t++) {
            if (t >= t0 && t <= t1) {  // (wave-uniform)
                const int sg = 64 * t + lane;
                if (sg >= base && sg < base + size) soff[t] = gstart + 2 * (sg - base);
            }
        }
        cur = base + size;
        // the lines this segment's image gains with the next chunk: points [8 k + 8 + JC, 8 k + 8 + 2 JC) behind its start
        const int nl = (JC >> 4) + 1;
        for (int i = 0; i < nl; i++) {
            const int e = pf_next + i;
            if (e < 128 && (e & 63) == lane) pf[e >> 6] = gstart + 8 * k + 8 + JC + 16 * i;
        }
        pf_next += nl;
    }
    const int T = (cur + 63) >> 6;
    tab_pf[(unit * 2 + 0) * 64 + lane] = pf[0];
    tab_pf[(unit * 2 + 1) * 64 + lane] = pf[1];
    int32_t *__restrict__ ts = tab_soff + ((size_t)unit * EDG_TMAX) * 64 + lane;
#pragma unroll
    for (int t = 0; t < EDG_TMAX; t++) ts[t * 64] = soff[t];
    tab_x2[unit * 64 + lane] = my_base + 4 * (lane - my_head);
    if (lane == 0) {
        cost[unit] = edg_unit_cost(m, JC, chunk_cost);
        tab_meta[unit] = make_int2(JC, T);
    }
}
// exclusive prefix sums in place, one workgroup (contiguous slice per thread); also clears the arrival counters and the queue
__global__ __launch_bounds__(1024) void kvm_ed_g_scan_kernel(int64_t *__restrict__ cost, int64_t units, unsigned *__restrict__ arrived,
                                                             unsigned *__restrict__ next_piece) {
    __shared__ int64_t wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = (units + 1023) / 1024;
    const int64_t b0 = min(units, (int64_t)tid * per), b1 = min(units, b0 + per);
    int64_t mine = 0;
    for (int64_t b = b0; b < b1; b++) mine += cost[b];
    int64_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int64_t run = inc - mine;
    for (int w = 0; w < wave; w++) run += wave_tot[w];
    for (int64_t b = b0; b < b1; b++) {
        const int64_t t = cost[b];
        cost[b] = run;
        run += t;
        arrived[b] = 0u;
    }
    if (tid == 1023) cost[units] = run;   // (the last thread's running sum is the total: slices are in thread order)
    if (tid == 0) *next_piece = 0u;
}

// scratch of the launch: partial accumulators of (units + pieces) passes | cost prefix sums | arrival counters + queue head |
// per-unit tables
#define EDG_PIECES_PER_WAVE 2
static int64_t edg_pieces(int64_t nw) {
    int64_t k = EDG_PIECES_PER_WAVE;
    if (const char *e = getenv("KVM_EDG_PIECES")) k = std::max<int64_t>(1, atoll(e));  // experiments
    return nw * k;
}
struct edg_layout {
    size_t parts, unit_off, arrived, tab_meta, tab_x2, tab_pf, tab_soff, total;
};
static edg_layout edg_scratch_layout(int64_t units, int64_t pieces) {
    edg_layout L;
    size_t o = 0;
    L.parts = o;     o += (size_t)(units + pieces + 1) * 8 * 64 * sizeof(double);
    L.unit_off = o;  o += (size_t)(units + 2) * sizeof(int64_t);
    L.arrived = o;   o += ((size_t)(units + 2) * sizeof(unsigned) + 15) / 16 * 16;
    L.tab_meta = o;  o += (size_t)units * sizeof(int2);
    L.tab_x2 = o;    o += (size_t)units * 64 * sizeof(int32_t);
    L.tab_pf = o;    o += (size_t)units * 2 * 64 * sizeof(int32_t);
    L.tab_soff = o;  o += (size_t)units * EDG_TMAX * 64 * sizeof(int32_t);
    L.total = o + 256;
    return L;
}
size_t kvm_ed_g_scratch_bytes(int64_t n_tiles8, int num_cu) {
    const int64_t units = (n_tiles8 + 63) / 64;
    return edg_scratch_layout(units, edg_pieces((int64_t)num_cu * 4 * EDG_WPS)).total;
}

hipError_t kvm_launch_ed_g(const double *series, const double *series_sh, const double2 *prefix, const kvm_query_dev &qd,
                           const kvm_tile *tiles8, int64_t n_tiles8, kvm_surv *surv, int64_t cap, unsigned long long *counters,
                           int num_cu, int32_t *out_off, double *out_dist, int64_t out_cap, void *scratch, hipStream_t s) {
    if (n_tiles8 <= 0) return hipSuccess;
    const int64_t units = (n_tiles8 + 63) / 64;
    const int gpu = (qd.m + 15) / 16;
    const int64_t total = units * gpu;
    // one wavefront per wave slot of the chip; short lists: at least 32 groups (512 points of 512 candidates) per piece
    const int64_t slots = (int64_t)num_cu * 4 * EDG_WPS;
    const int64_t pieces = std::max<int64_t>(1, std::min<int64_t>(edg_pieces(slots), total / 32));
    const int64_t nw = std::min<int64_t>(slots, pieces);
    const edg_layout L = edg_scratch_layout(units, edg_pieces(slots));
    char *base = (char *)scratch;
    edg_args A;
    A.series = series; A.series_sh = series_sh; A.prefix = prefix; A.qd = qd; A.surv = surv; A.cap = cap; A.counters = counters;
    A.out_off = out_off; A.out_dist = out_dist; A.out_cap = out_cap;
    A.parts = (double *)(base + L.parts);
    int64_t *unit_off = (int64_t *)(base + L.unit_off);
    A.unit_off = unit_off;
    A.arrived = (unsigned *)(base + L.arrived);
    A.next_piece = A.arrived + units;
    A.tab_meta = (const int2 *)(base + L.tab_meta);
    A.tab_x2 = (const int32_t *)(base + L.tab_x2);
    A.tab_soff = (const int32_t *)(base + L.tab_soff);
    A.tab_pf = (const int32_t *)(base + L.tab_pf);
    A.units = units; A.groups_per_unit = gpu; A.n_waves = (int)nw; A.n_pieces = pieces;
    A.dbg = getenv("KVM_EDG_DBG") ? atoi(getenv("KVM_EDG_DBG")) : 0;
    static unsigned long long *d_phase = nullptr;
    A.phase_t = nullptr;
    if (A.dbg & 2) {
        if (!d_phase) (void)hipMalloc((void **)&d_phase, 64);
        (void)hipMemsetAsync(d_phase, 0, 64, s);
        A.phase_t = d_phase;
    }
    hipLaunchKernelGGL(kvm_ed_g_plan_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, tiles8, n_tiles8, units, (int)qd.m,
                       getenv("KVM_EDG_CHUNK_COST") ? atoi(getenv("KVM_EDG_CHUNK_COST")) : EDG_CHUNK_COST, unit_off,
                       (int32_t *)(base + L.tab_soff), (int32_t *)(base + L.tab_x2), (int2 *)(base + L.tab_meta), (int32_t *)(base + L.tab_pf));
    hipLaunchKernelGGL(kvm_ed_g_scan_kernel, dim3(1), dim3(1024), 0, s, unit_off, units, A.arrived, A.next_piece);
    if (getenv("KVM_TRACE"))
        fprintf(stderr, "[kvm] ed_g m %d tiles8 %lld units %lld waves %lld pieces %lld\n", (int)qd.m, (long long)n_tiles8, (long long)units,
                (long long)nw, (long long)pieces);
    hipLaunchKernelGGL(kvm_ed_g_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, s, A, tiles8, n_tiles8);
    if (A.phase_t) {  // experiments (blocking): where the wavefronts' time goes, per wavefront on average
        unsigned long long h[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, d_phase, 64, hipMemcpyDeviceToHost);
        fprintf(stderr, "[kvm] ed_g per wavefront (us): tables %.1f  chunk loops %.1f  hand-over + decision %.1f  queue %.1f  | alive %.1f  | %.1f passes\n",
                h[0] * 0.01 / nw, h[1] * 0.01 / nw, h[2] * 0.01 / nw, h[3] * 0.01 / nw, h[5] * 0.01 / nw, (double)h[4] / nw);
    }
    return hipGetLastError();
}
