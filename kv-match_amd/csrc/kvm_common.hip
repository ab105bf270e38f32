// kvm_common.hip -- staging kernels: big-endian decode, block-restarted prefix sums, sliding envelope.
// gfx950 (CDNA4) only.  All three are single-pass HBM-bound streaming kernels.
#include "kvm_internal.h"


// ---------------------------------------------------------------------------------------------------
// The phase-2 plan on the device: interval list -> runs (clipped start, count, exclusive prefix sums of tiles and
// windows), compacted to the non-empty ones, in list order.  One thread per interval, two launches:
//   kvm_plan_totals_kernel  the four sums of every 256-interval block;
//   kvm_plan_runs_kernel    every block adds up the totals of the blocks before it, scans its own 256 items and writes
//                           its runs.
// (A single workgroup doing all of it took 43-61 us at k = 15 675: ~80 instructions of 64-bit integer arithmetic per
// interval on one CU.)  The host sums the same per-interval rule (kvm_plan_interval) for the launch geometry only.
#define PLAN_THREADS 256
struct plan_sums {
    long long nr, ta, tb, tc, w;
};
static_assert(sizeof(plan_sums) == KVM_PLAN_SUMS_BYTES, "kvm_verify.hip sizes the block-total scratch of kvm_launch_plan_runs with it");
__device__ __forceinline__ plan_sums plan_add(plan_sums a, plan_sums b) {
    return plan_sums{a.nr + b.nr, a.ta + b.ta, a.tb + b.tb, a.tc + b.tc, a.w + b.w};
}
__device__ __forceinline__ plan_sums plan_sub(plan_sums a, plan_sums b) {
    return plan_sums{a.nr - b.nr, a.ta - b.ta, a.tb - b.tb, a.tc - b.tc, a.w - b.w};
}
__device__ __forceinline__ plan_sums plan_shfl_xor(plan_sums v, int o) {
    return plan_sums{__shfl_xor(v.nr, o, 64), __shfl_xor(v.ta, o, 64), __shfl_xor(v.tb, o, 64), __shfl_xor(v.tc, o, 64),
                     __shfl_xor(v.w, o, 64)};
}
__device__ __forceinline__ plan_sums plan_shfl_up(plan_sums v, int o) {
    return plan_sums{__shfl_up(v.nr, o, 64), __shfl_up(v.ta, o, 64), __shfl_up(v.tb, o, 64), __shfl_up(v.tc, o, 64),
                     __shfl_up(v.w, o, 64)};
}
// sum over the workgroup, result in every thread
__device__ __forceinline__ plan_sums plan_block_sum(plan_sums v, plan_sums *wave_tot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = plan_add(v, plan_shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = v;
    __syncthreads();
    plan_sums t = wave_tot[0];
    for (int w = 1; w < PLAN_THREADS / 64; w++) t = plan_add(t, wave_tot[w]);
    __syncthreads();
    return t;
}
__device__ __forceinline__ plan_sums plan_item_sums(const kvm_plan_item &it) {
    return plan_sums{it.c > 0 ? 1 : 0, (long long)it.tiles_a, (long long)it.tiles_b, (long long)it.tiles_c, (long long)it.c};
}

// `src_left` / `src_right` may be the pinned host staging buffer itself (zero-copy over PCIe: 8 B per interval, no copy
// command and no copy-engine signal in front of the kernel); the kernel leaves a device copy in `left` / `right` for
// kvm_plan_runs_kernel.
__global__ __launch_bounds__(PLAN_THREADS) void kvm_plan_totals_kernel(const int32_t *__restrict__ src_left,
                                                                      const int32_t *__restrict__ src_right,
                                                                      int32_t *__restrict__ left, int32_t *__restrict__ right,
                                                                      int32_t k, int32_t shift, int64_t hi_max, int hybrid,
                                                                      int tile_log2, plan_sums *__restrict__ block_tot) {
    __shared__ plan_sums wave_tot[PLAN_THREADS / 64];
    const int64_t p = (int64_t)blockIdx.x * PLAN_THREADS + threadIdx.x;
    plan_sums mine{0, 0, 0, 0, 0};
    if (p < k) {
        const int32_t l = src_left[p], r = src_right[p];
        left[p] = l;
        right[p] = r;
        mine = plan_item_sums(kvm_plan_interval(l, r, shift, hi_max, hybrid, tile_log2));
    }
    const plan_sums t = plan_block_sum(mine, wave_tot);
    if (threadIdx.x == 0) block_tot[blockIdx.x] = t;
}

__global__ __launch_bounds__(PLAN_THREADS) void kvm_plan_runs_kernel(const int32_t *__restrict__ left, const int32_t *__restrict__ right,
                                                                    int32_t k, int32_t shift, int64_t hi_max, int hybrid,
                                                                    int tile_log2, const plan_sums *__restrict__ block_tot,
                                                                    int scanned, kvm_run *__restrict__ runs,
                                                                    const int32_t *__restrict__ chunk_lo,
                                                                    const int32_t *__restrict__ chunk_hi,
                                                                    int2 *__restrict__ run_chunk) {
    __shared__ plan_sums wave_tot[PLAN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // totals of the blocks before this one: summed here (few blocks), or already scanned by kvm_plan_scan_kernel
    plan_sums before{0, 0, 0, 0, 0};
    if (scanned) {
        before = block_tot[blockIdx.x];
    } else {
        for (int64_t b = tid; b < (int64_t)blockIdx.x; b += PLAN_THREADS) before = plan_add(before, block_tot[b]);
        before = plan_block_sum(before, wave_tot);
    }
    const int64_t p = (int64_t)blockIdx.x * PLAN_THREADS + tid;
    kvm_plan_item it;
    it.lo = 1; it.c64 = 0; it.c = 0; it.tiles_a = 0; it.tiles_b = 0; it.tiles_c = 0;
    if (p < k) it = kvm_plan_interval(left[p], right[p], shift, hi_max, hybrid, tile_log2);
    const plan_sums mine = plan_item_sums(it);
    plan_sums inc = mine;  // inclusive scan inside the wave, then across the wave totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const plan_sums t = plan_shfl_up(inc, o);
        if (lane >= o) inc = plan_add(inc, t);
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    plan_sums off = before;
    for (int w = 0; w < wave; w++) off = plan_add(off, wave_tot[w]);
    if (it.c > 0) {
        kvm_run r;
        r.start = (int32_t)it.lo;  // the reference restarts ex/ex2 here (K/NormQueryEngine.java:487-491)
        r.count = (int32_t)it.c;
        r.tile_base = off.ta + inc.ta - mine.ta;
        r.tile_base_b = it.tiles_b ? off.tb + inc.tb - mine.tb : -1;
        r.tile_base_c = it.tiles_c ? off.tc + inc.tc - mine.tc : -1;
        r.win_base = off.w + inc.w - mine.w;
        runs[off.nr + inc.nr - mine.nr] = r;
        if (run_chunk) run_chunk[off.nr + inc.nr - mine.nr] = make_int2(chunk_lo[p], chunk_hi[p]);
    }
}

// Long lists (more than PLAN_SCAN_ABOVE blocks): the block totals are turned into exclusive prefixes in place by one
// workgroup (contiguous slice per thread, block scan of the slice sums) so that kvm_plan_runs_kernel does not sum
// O(blocks) totals per block.
#define PLAN_SCAN_ABOVE 1024
__global__ __launch_bounds__(1024) void kvm_plan_scan_kernel(plan_sums *__restrict__ block_tot, int64_t blocks) {
    __shared__ plan_sums wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = (blocks + 1023) / 1024;
    const int64_t b0 = min(blocks, (int64_t)tid * per), b1 = min(blocks, b0 + per);
    plan_sums mine{0, 0, 0, 0, 0};
    for (int64_t b = b0; b < b1; b++) mine = plan_add(mine, block_tot[b]);
    plan_sums inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const plan_sums t = plan_shfl_up(inc, o);
        if (lane >= o) inc = plan_add(inc, t);
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    plan_sums run{0, 0, 0, 0, 0};
    for (int w = 0; w < wave; w++) run = plan_add(run, wave_tot[w]);
    run = plan_add(run, plan_sub(inc, mine));  // exclusive inside the wave
    for (int64_t b = b0; b < b1; b++) {
        const plan_sums t = block_tot[b];
        block_tot[b] = run;
        run = plan_add(run, t);
    }
}

// `scratch`: KVM_PLAN_SUMS_BYTES per 256 intervals (block totals)
hipError_t kvm_launch_plan_runs(const int32_t *src_left, const int32_t *src_right, int32_t *left, int32_t *right, int32_t k,
                                int32_t shift, int64_t hi_max, int hybrid, int tile_log2, kvm_run *runs, void *scratch,
                                hipStream_t s, const int32_t *chunk_lo, const int32_t *chunk_hi, int2 *run_chunk) {
    if (k <= 0) return hipSuccess;
    const int64_t blocks = ((int64_t)k + PLAN_THREADS - 1) / PLAN_THREADS;
    plan_sums *bt = reinterpret_cast<plan_sums *>(scratch);
    hipLaunchKernelGGL(kvm_plan_totals_kernel, dim3((unsigned)blocks), dim3(PLAN_THREADS), 0, s, src_left, src_right, left, right,
                       k, shift, hi_max, hybrid, tile_log2, bt);
    const int scanned = blocks > PLAN_SCAN_ABOVE;
    if (scanned) hipLaunchKernelGGL(kvm_plan_scan_kernel, dim3(1), dim3(1024), 0, s, bt, blocks);
    hipLaunchKernelGGL(kvm_plan_runs_kernel, dim3((unsigned)blocks), dim3(PLAN_THREADS), 0, s, left, right, k, shift, hi_max,
                       hybrid, tile_log2, bt, scanned, runs, chunk_lo, chunk_hi, run_chunk);
    return hipGetLastError();
}

// two device words set in stream order (count + work-queue head of a follow-up DP launch)
__global__ void kvm_set_u64_kernel(unsigned long long *a, unsigned long long va, unsigned long long *b, unsigned long long vb) {
    if (threadIdx.x == 0) {
        *a = va;
        if (b) *b = vb;
    }
}
hipError_t kvm_launch_set_u64(unsigned long long *a, unsigned long long va, unsigned long long *b, unsigned long long vb,
                              hipStream_t s) {
    hipLaunchKernelGGL(kvm_set_u64_kernel, dim3(1), dim3(64), 0, s, a, va, b, vb);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Big-endian fp64 -> native.  Replaces Bytes.toDouble per value (K/common/entity/TimeSeriesNode.java:46-52).
// 16 B per lane, grid-stride; in place.
__global__ __launch_bounds__(256) void kvm_bswap64_kernel(uint64_t *__restrict__ d, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    int64_t stride = (int64_t)gridDim.x * blockDim.x * 2;
    for (; i + 1 < n; i += stride) {
        ulonglong2 v = *reinterpret_cast<ulonglong2 *>(d + i);
        v.x = __builtin_bswap64(v.x);
        v.y = __builtin_bswap64(v.y);
        *reinterpret_cast<ulonglong2 *>(d + i) = v;
    }
    if (i < n) d[i] = __builtin_bswap64(d[i]);  // odd tail (at most one thread hits this)
}

hipError_t kvm_launch_bswap64(uint64_t *d, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int64_t pairs = (n + 1) / 2;
    int64_t blocks = (pairs + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(kvm_bswap64_kernel, dim3((unsigned)blocks), dim3(256), 0, s, d, n);
    return hipGetLastError();
}


// Grid of the tile-expansion kernels: one wavefront per run -- but the launch also carries the call's query block over from
// the pinned staging buffer (grid-stride loop at its head), and a list of a few runs (a selective query, a full scan's single
// run) must not leave a 200 KB block to one workgroup's ~100 dependent PCIe round trips per thread: at least as many
// workgroups as the dedicated upload kernel used (one per 2 KB, up to 512).  Workgroups beyond the runs leave after the copy.
static unsigned kvm_expand_grid(int64_t n_runs, size_t up_bytes) {
    const int64_t by_runs = (n_runs + 3) / 4, by_bytes = (int64_t)std::min<size_t>((up_bytes / 8 + 255) / 256, 512);
    return (unsigned)std::max<int64_t>(by_runs, std::max<int64_t>(by_bytes, 1));
}

// Hybrid ED tiling (kvm_ed_dot.hip): per run floor(count/64) full 64-candidate tiles; what is left goes to class A, to
// class B and / or to class C by kvm_ed_remainder_classes (the class-B tile takes the first <= 32 candidates of the
// remainder, the class-C tile the rest).
__global__ __launch_bounds__(256) void kvm_expand_tiles_hybrid_kernel(const kvm_run *__restrict__ runs, int64_t n_runs,
                                                                      int hybrid, kvm_tile *__restrict__ tilesA,
                                                                      kvm_tile *__restrict__ tilesB,
                                                                      kvm_tile *__restrict__ tilesC,
                                                                      unsigned long long *__restrict__ counters,
                                                                      int32_t *__restrict__ run_maxk,
                                                                      unsigned long long *__restrict__ up_dst,
                                                                      const unsigned long long *__restrict__ up_src, int64_t up_n8) {
    if (blockIdx.x == 0 && threadIdx.x < KVM_CNT_TOTAL) counters[threadIdx.x] = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < up_n8; i += (int64_t)gridDim.x * blockDim.x) up_dst[i] = up_src[i];  // (the call's query block: see kvm_expand_tiles_kernel)
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_runs) return;
    const kvm_run run = runs[r];
    if (run_maxk && lane == 0) run_maxk[r] = 0;
    const int full = run.count / 64, rem = run.count % 64;
    uint32_t ra, rb, rc;
    kvm_ed_remainder_classes((uint32_t)rem, ra, rb, rc);
    if (hybrid == 1) {  // no class C for this list (kvm_plan_tiles): a remainder of <= 16 is a class-B tile
        rb += rc;
        rc = 0;
    }
    const int nA = full + (int)ra;
    for (int t = lane; t < nA; t += 64) {
        kvm_tile tl;
        tl.start = run.start + t * 64;
        tl.count = min(64, run.count - t * 64);
        tl.run_start = run.start;
        tl.run = (int32_t)r;
        tilesA[run.tile_base + t] = tl;
    }
    const int nb = rb ? min(rem, (int)KVM_EDB_TILE) : 0;  // candidates of the class-B tile
    if (lane == 0 && rb) {
        kvm_tile tl;
        tl.start = run.start + full * 64;
        tl.count = nb;
        tl.run_start = run.start;
        tl.run = (int32_t)r;
        tilesB[run.tile_base_b] = tl;
    }
    if (lane == 1 && rc) {
        kvm_tile tl;
        tl.start = run.start + full * 64 + nb;
        tl.count = rem - nb;
        tl.run_start = run.start;
        tl.run = (int32_t)r;
        tilesC[run.tile_base_c] = tl;
    }
}

hipError_t kvm_launch_expand_tiles_hybrid(const kvm_run *runs, int64_t n_runs, int hybrid, kvm_tile *tilesA, kvm_tile *tilesB,
                                          kvm_tile *tilesC, unsigned long long *counters, int32_t *run_maxk, hipStream_t s,
                                          void *up_dst, const void *up_src_pinned, size_t up_bytes) {
    if (n_runs <= 0) return up_bytes ? kvm_launch_upload(up_dst, up_src_pinned, up_bytes, s) : hipSuccess;
    hipLaunchKernelGGL(kvm_expand_tiles_hybrid_kernel, dim3(kvm_expand_grid(n_runs, up_bytes)), dim3(256), 0, s, runs, n_runs,
                       hybrid, tilesA, tilesB, tilesC, counters, run_maxk, (unsigned long long *)up_dst,
                       (const unsigned long long *)up_src_pinned, (int64_t)((up_bytes + 7) / 8));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// [count | first pack_cap (offset, distance) pairs] as fp64 pairs: the payload of the multi-GPU all-gather.
__global__ __launch_bounds__(256) void kvm_pack_answers_kernel(const int32_t *__restrict__ off,
                                                               const double *__restrict__ dist,
                                                               const int64_t *__restrict__ count, int64_t pack_cap,
                                                               double2 *__restrict__ packed) {
    const int64_t cnt = *count;
    const int64_t take = cnt < pack_cap ? cnt : pack_cap;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= pack_cap; i += (int64_t)gridDim.x * blockDim.x) {
        if (i == 0)
            packed[0] = make_double2((double)cnt, 0.0);
        else if (i <= take)
            packed[i] = make_double2((double)off[i - 1], dist[i - 1]);
        // (what lies behind the pairs present is not written: nothing reads it -- the readback and the merge go by the count)
    }
}

// *dst = src ? *src : 0 -- one word moved by a kernel instead of a copy command (a device-to-device hipMemcpyAsync goes through
// the runtime's blit manager: ~10 us of host time per call under a lock every stream of the device shares)
__global__ void kvm_copy_u64_kernel(unsigned long long *__restrict__ dst, const unsigned long long *__restrict__ src) {
    if (threadIdx.x == 0) *dst = src ? *src : 0ULL;
}
hipError_t kvm_launch_copy_u64(unsigned long long *dst, const unsigned long long *src, hipStream_t s) {
    hipLaunchKernelGGL(kvm_copy_u64_kernel, dim3(1), dim3(64), 0, s, dst, src);
    return hipGetLastError();
}

// Result readback without a copy command: one small block stores the counters and the first `spec` answers into
// device-mapped pinned host memory; the host only has to wait for the stream.
__global__ __launch_bounds__(256) void kvm_readback_kernel(const unsigned long long *__restrict__ counters,
                                                           const int32_t *__restrict__ off,
                                                           const double *__restrict__ dist, int64_t spec,
                                                           unsigned long long *__restrict__ host) {
    const int t = threadIdx.x;
    if (t < KVM_CNT_TOTAL) host[t] = counters[t];
    const int64_t cnt = (int64_t)counters[KVM_CNT_ANSWERS];
    const int64_t take = cnt < spec ? cnt : spec;
    int32_t *hoff = reinterpret_cast<int32_t *>(host + KVM_CNT_TOTAL);
    double *hdist = reinterpret_cast<double *>(hoff + ((spec + 1) & ~(int64_t)1));  // 8-byte aligned
    for (int64_t i = t; i < take; i += 256) {
        hoff[i] = off[i];
        hdist[i] = dist[i];
    }
}

hipError_t kvm_launch_readback(const unsigned long long *counters, const int32_t *off, const double *dist, int64_t spec,
                               void *host, hipStream_t s) {
    hipLaunchKernelGGL(kvm_readback_kernel, dim3(1), dim3(256), 0, s, counters, off, dist, spec,
                       (unsigned long long *)host);
    return hipGetLastError();
}

// Small host -> device uploads without a copy command: the source is the pinned staging buffer, read over PCIe with
// 8-byte loads (a copy command costs its own ~8 us plus ~10 us of copy-engine signalling before the next kernel starts).
__global__ __launch_bounds__(256) void kvm_upload_kernel(unsigned long long *__restrict__ dst, const unsigned long long *__restrict__ src,
                                                         int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// (offsets inside the staging buffer are multiples of 8; a length that is not is rounded up into the block's padding)
hipError_t kvm_launch_upload(void *dst, const void *src_pinned, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    const int64_t n8 = (int64_t)((bytes + 7) / 8);
    const unsigned blocks = (unsigned)std::min<int64_t>((n8 + 255) / 256, 512);
    hipLaunchKernelGGL(kvm_upload_kernel, dim3(blocks), dim3(256), 0, s, (unsigned long long *)dst,
                       (const unsigned long long *)src_pinned, n8);
    return hipGetLastError();
}

hipError_t kvm_launch_pack_answers(const int32_t *off, const double *dist, const int64_t *count, int64_t pack_cap,
                                   double2 *packed, hipStream_t s) {
    hipLaunchKernelGGL(kvm_pack_answers_kernel, dim3((unsigned)((pack_cap + 256) / 256)), dim3(256), 0, s, off, dist, count,
                       pack_cap, packed);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Runs -> tiles.  The host clips the phase-1 intervals into runs (K/QueryEngine.java:346-354) and knows each run's
// first tile index; one wavefront per run writes its tile descriptors (16 B each, coalesced).
__global__ __launch_bounds__(256) void kvm_expand_tiles_kernel(const kvm_run *__restrict__ runs, int64_t n_runs,
                                                               int32_t tile_max, kvm_tile *__restrict__ tiles,
                                                               unsigned long long *__restrict__ counters,
                                                               int32_t *__restrict__ run_maxk,
                                                               const double *__restrict__ blockmax, double *__restrict__ run_hist,
                                                               int32_t m, int64_t n, unsigned long long *__restrict__ up_dst,
                                                               const unsigned long long *__restrict__ up_src, int64_t up_n8) {
    if (blockIdx.x == 0 && threadIdx.x < KVM_CNT_TOTAL) counters[threadIdx.x] = 0;  // (saves a fill launch per call)
    // ... and the upload launch: the call's query block comes over from the pinned staging buffer here (kvm_upload_kernel's
    // loop; nothing between the plan kernels and this one reads it)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < up_n8; i += (int64_t)gridDim.x * blockDim.x) up_dst[i] = up_src[i];
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_runs) return;
    const kvm_run run = runs[r];
    if (run_maxk && lane == 0) run_maxk[r] = 0;  // cNSM: clear the replay scratch of this run (windows-to-replay count)
    const int nt = (run.count + tile_max - 1) / tile_max;
    for (int t = lane; t < nt; t += 64) {
        kvm_tile tl;
        tl.start = run.start + t * tile_max;
        tl.count = min(tile_max, run.count - t * tile_max);
        tl.run_start = run.start;
        tl.run = (int32_t)r;
        tiles[run.tile_base + t] = tl;
    }
    if (run_hist) {
        // cNSM: the largest block maximum among the prefix blocks the run's data range [start, start + count + m - 2]
        // (1-based points) touches -- what the reference's running sums of that interval pass over, rounded up to whole
        // blocks (kvm_hist_*).  An ordinary run touches one or two blocks, the run of a full scan thousands.
        const int64_t lo = ((int64_t)run.start - 1) >> KVM_PS_BLOCK_LOG2;
        int64_t last = (int64_t)run.start - 1 + run.count + m - 2;
        if (last > n - 1) last = n - 1;
        const int64_t hi = last >> KVM_PS_BLOCK_LOG2;
        double mx = 0.0;
        for (int64_t b = lo + lane; b <= hi; b += 64) {
            const double t = blockmax[b];
            mx = (mx != mx || t != t) ? __builtin_nan("") : fmax(mx, t);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double t = __shfl_xor(mx, o, 64);
            mx = (mx != mx || t != t) ? __builtin_nan("") : fmax(mx, t);
        }
        if (lane == 0) run_hist[r] = mx;
    }
}

hipError_t kvm_launch_expand_tiles(const kvm_run *runs, int64_t n_runs, int32_t tile_max, kvm_tile *tiles,
                                   unsigned long long *counters, int32_t *run_maxk, hipStream_t s, const double *blockmax,
                                   double *run_hist, int32_t m, int64_t n, void *up_dst, const void *up_src_pinned, size_t up_bytes) {
    if (n_runs <= 0) return up_bytes ? kvm_launch_upload(up_dst, up_src_pinned, up_bytes, s) : hipSuccess;
    hipLaunchKernelGGL(kvm_expand_tiles_kernel, dim3(kvm_expand_grid(n_runs, up_bytes)), dim3(256), 0, s, runs, n_runs,
                       tile_max, tiles, counters, run_maxk, blockmax, run_hist, m, n, (unsigned long long *)up_dst,
                       (const unsigned long long *)up_src_pinned, (int64_t)((up_bytes + 7) / 8));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Block-restarted inclusive prefix sums of (x - c_b) and (x - c_b)^2, c_b = first point of the block
// (see KVM_PS_BLOCK in kvm_internal.h).
// One workgroup scans one block of 16384 points as 16 sub-tiles of 1024 (4 consecutive points per
// thread: 32 B loads, 64 B stores per lane).  Blocks are independent, so this is a one-pass kernel:
// 8 B read + 16 B written per point.
__global__ __launch_bounds__(256) void kvm_prefix_kernel(const double *__restrict__ x, double2 *__restrict__ P,
                                                         int64_t n) {
    __shared__ double wtot1[4], wtot2[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x << KVM_PS_BLOCK_LOG2;
    double carry1 = 0.0, carry2 = 0.0;
    const double shift = x[base];  // c_b
    for (int sub = 0; sub < KVM_PS_BLOCK / 1024; sub++) {
        const int64_t i0 = base + sub * 1024 + tid * 4;
        if (base + sub * 1024 >= n) break;  // uniform across the workgroup
        double v[4];
        if (i0 + 3 < n) {
            const double4 t = *reinterpret_cast<const double4 *>(x + i0);
            v[0] = t.x - shift; v[1] = t.y - shift; v[2] = t.z - shift; v[3] = t.w - shift;
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] = (i0 + c < n) ? x[i0 + c] - shift : 0.0;
        }
        double a1[4], a2[4];
        a1[0] = v[0];
        a2[0] = v[0] * v[0];
#pragma unroll
        for (int c = 1; c < 4; c++) {
            a1[c] = a1[c - 1] + v[c];
            a2[c] = a2[c - 1] + v[c] * v[c];
        }
        // wave-inclusive scan of the per-thread totals
        double s1 = a1[3], s2 = a2[3];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            double t1 = __shfl_up(s1, o, 64), t2 = __shfl_up(s2, o, 64);
            if (lane >= o) {
                s1 += t1;
                s2 += t2;
            }
        }
        if (lane == 63) {
            wtot1[wave] = s1;
            wtot2[wave] = s2;
        }
        __syncthreads();
        double off1 = carry1, off2 = carry2;
        for (int w = 0; w < wave; w++) {
            off1 += wtot1[w];
            off2 += wtot2[w];
        }
        // exclusive prefix of this thread = the inclusive scan of the lane before it.  (NOT `inclusive - own total`: an
        // entry must depend on the points before it only -- with a large value in the thread's own group, e.g. the first
        // point of a burst 1e150 times the local scale, the subtraction wiped out the prefix of the three points in front
        // of it and the error bounds of kvm_prefix_err, which follow the magnitude of the ENTRY, no longer held.)
        const double e1 = __shfl_up(s1, 1, 64), e2 = __shfl_up(s2, 1, 64);
        off1 += lane ? e1 : 0.0;
        off2 += lane ? e2 : 0.0;
        if (i0 + 3 < n) {
#pragma unroll
            for (int c = 0; c < 4; c++) P[i0 + c] = make_double2(off1 + a1[c], off2 + a2[c]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (i0 + c < n) P[i0 + c] = make_double2(off1 + a1[c], off2 + a2[c]);
        }
        carry1 += (wtot1[0] + wtot1[1]) + (wtot1[2] + wtot1[3]);
        carry2 += (wtot2[0] + wtot2[1]) + (wtot2[2] + wtot2[3]);
        __syncthreads();
    }
}

// max x^2 of every prefix block; NaN if the block holds a NaN (fmax would drop it).  One workgroup per block.
__global__ __launch_bounds__(256) void kvm_blockmax_kernel(const double *__restrict__ x, int64_t n, double *__restrict__ out) {
    __shared__ double red[4];
    const int64_t b0 = (int64_t)blockIdx.x << KVM_PS_BLOCK_LOG2;
    double mx = 0.0;
    bool nan = false;
    for (int64_t i = b0 + threadIdx.x; i < b0 + KVM_PS_BLOCK && i < n; i += 256) {
        const double v = x[i];
        nan |= v != v;
        mx = fmax(mx, v * v);
    }
    if (nan) mx = __builtin_nan("");
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double t = __shfl_xor(mx, o, 64);
        mx = (mx != mx || t != t) ? __builtin_nan("") : fmax(mx, t);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = red[0];
        for (int w = 1; w < 4; w++) r = (r != r || red[w] != red[w]) ? __builtin_nan("") : fmax(r, red[w]);
        out[blockIdx.x] = r;
    }
}
hipError_t kvm_launch_blockmax(const double *x, int64_t n, double *blockmax, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int64_t blocks = (n + KVM_PS_BLOCK - 1) >> KVM_PS_BLOCK_LOG2;
    hipLaunchKernelGGL(kvm_blockmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, blockmax);
    return hipGetLastError();
}
hipError_t kvm_launch_prefix(const double *x, double2 *prefix, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + KVM_PS_BLOCK - 1) >> KVM_PS_BLOCK_LOG2;
    hipLaunchKernelGGL(kvm_prefix_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, prefix, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Sliding window means from the shifted prefix sums (index build support): one thread per window.
__global__ __launch_bounds__(256) void kvm_window_means_kernel(const double *__restrict__ x,
                                                               const double2 *__restrict__ P, int32_t w, int64_t first0,
                                                               int64_t count, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double s1, s2, c;
    kvm_window_shifted_sums(P, x, first0 + i, w, s1, s2, c);
    out[i] = c + s1 / w;
}

hipError_t kvm_launch_window_means(const double *x, const double2 *prefix, int32_t w, int64_t first0, int64_t count,
                                   double *out, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(kvm_window_means_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, x, prefix, w, first0,
                       count, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Sliding envelope over the whole series: u[i] = max x[max(0,i-r) .. min(n-1,i+r)], l[i] = min ... .
// Same values as DtwUtils.lowerUpperLemire (K/utils/DtwUtils.java:93-134) applied to the WHOLE series
// (pure selections, no arithmetic).  The reference applies it per fetched chunk, so its envelope is
// clipped at chunk ends; the global envelope is equal inside and looser at those ends -- still a valid
// bound, affects pruning only.
//
// Each workgroup produces ENV_TS (2048 or 1024) outputs from an LDS tile of ENV_TS + 2r points using the two-window
// (sparse-table) trick: with p = floor(log2(w)), w = 2r+1, max over a w-window = max(M_p[a], M_p[a+w-2^p])
// where M_p[i] = max x[i .. i+2^p).  M_p is built by p ping-pong doubling passes in LDS.
extern __shared__ double kvm_env_lds[];

template <int ENV_TS>
__global__ __launch_bounds__(256) void kvm_envelope_kernel(const double *__restrict__ x, int64_t n, int32_t r,
                                                           double *__restrict__ u, double *__restrict__ l) {
    const int tid = threadIdx.x;
    const int len = ENV_TS + 2 * r;  // tile incl. halo
    double *A = kvm_env_lds;         // ping
    double *B = kvm_env_lds + len;   // pong
    const int64_t t0 = (int64_t)blockIdx.x * ENV_TS;  // first output index
    const int64_t g0 = t0 - r;                        // global index of tile[0]
    const int w = 2 * r + 1;
    int p = 0;
    while ((2 << p) <= w) p++;  // 2^p <= w < 2^(p+1)

    for (int pass = 0; pass < 2; pass++) {  // pass 0: max, pass 1: min
        const double pad = pass == 0 ? -__builtin_huge_val() : __builtin_huge_val();
        for (int i = tid; i < len; i += 256) {
            int64_t g = g0 + i;
            A[i] = (g >= 0 && g < n) ? x[g] : pad;
        }
        __syncthreads();
        double *src = A, *dst = B;
        for (int t = 0; t < p; t++) {
            const int h = 1 << t;
            for (int i = tid; i < len; i += 256) {
                double a = src[i];
                double b = (i + h < len) ? src[i + h] : pad;
                dst[i] = pass == 0 ? fmax(a, b) : fmin(a, b);
            }
            __syncthreads();
            double *tmp = src;
            src = dst;
            dst = tmp;
        }
        const int span = 1 << p;
        for (int i = tid; i < ENV_TS; i += 256) {
            int64_t g = t0 + i;
            if (g < n) {
                // window = tile indices [i, i+w)
                double a = src[i], b = src[i + w - span];
                if (pass == 0)
                    u[g] = fmax(a, b);
                else
                    l[g] = fmin(a, b);
            }
        }
        __syncthreads();
    }
}

// Fallback for very wide bands (LDS tile would not fit): direct O(w) scan per point.
__global__ __launch_bounds__(256) void kvm_envelope_naive_kernel(const double *__restrict__ x, int64_t n, int32_t r,
                                                                 double *__restrict__ u, double *__restrict__ l) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t lo = i - r < 0 ? 0 : i - r, hi = i + r > n - 1 ? n - 1 : i + r;
    double mx = x[lo], mn = x[lo];
    for (int64_t g = lo + 1; g <= hi; g++) {
        double v = x[g];
        mx = fmax(mx, v);
        mn = fmin(mn, v);
    }
    u[i] = mx;
    l[i] = mn;
}

hipError_t kvm_launch_envelope(const double *x, int64_t n, int32_t rho, double *u, double *l, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    // output tile of 2048 points while the tile + halo fits 64 KiB of LDS (rho <= 1024); wider bands use 1024-point
    // tiles and up to the full 160 KiB of the CU (rho <= 4608); beyond that the O(w)-per-point fallback
    const size_t lds_big = (size_t)(2048 + 2 * (int64_t)rho) * 2 * sizeof(double);
    const size_t lds_small = (size_t)(1024 + 2 * (int64_t)rho) * 2 * sizeof(double);
    if (lds_big <= 64 * 1024) {
        int64_t blocks = (n + 2047) / 2048;
        hipLaunchKernelGGL(kvm_envelope_kernel<2048>, dim3((unsigned)blocks), dim3(256), lds_big, s, x, n, rho, u, l);
    } else if (lds_small <= 160 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kvm_envelope_kernel<1024>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small);
        if (e != hipSuccess) return e;
        int64_t blocks = (n + 1023) / 1024;
        hipLaunchKernelGGL(kvm_envelope_kernel<1024>, dim3((unsigned)blocks), dim3(256), lds_small, s, x, n, rho, u, l);
    } else {
        int64_t blocks = (n + 255) / 256;
        hipLaunchKernelGGL(kvm_envelope_naive_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, rho, u, l);
    }
    return hipGetLastError();
}
