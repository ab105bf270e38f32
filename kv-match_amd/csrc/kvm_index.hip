// kvm_index.hip -- index build, step 1 ("fixed-width index rows") of K/IndexBuilder.java:186-300 for one window
// length w, bit for bit.
//
// The reference streams the series in buffers of EPOCH = 100000 points, restarts its running sum at every buffer and,
// per window position, forms  mean = ex / w  and the row key  toRound(mean)  (K/utils/MeanIntervalUtils.java:50-60);
// consecutive positions with the same key extend one interval of that key's row, cut every 255 positions.  The key is
// a rounding of the mean to 0.05, so a mean computed by any other arithmetic (prefix sums, kvm_window_means) lands on
// the other side of a rounding boundary every ~1e12-th position -- rarely, but an index is supposed to be THE index.
// So the running sum is replayed exactly:
//   kvm_index_ex_kernel   one workgroup per EPOCH buffer (they are independent: n = 1e8 -> 1001 of them): lane 0 of
//                         wave 0 runs `ex += d; [record ex]; ex -= oldest` strictly in order, from LDS chunks that the
//                         other three waves stage (same producer / consumer scheme as the running-sum replay of the
//                         exact pass, kvm_exact.hip).  Two dependent additions per position: ~20 cycles, 1 ms a buffer.
//   kvm_index_keys_kernel mean = ex / w and the key's bit pattern for every position, in parallel (the division does
//                         not belong in the one-lane loop).
//   hipcub run-length encode of the key bit patterns (Double.equals semantics: NaN == NaN, 0.0 != -0.0), a plain
//                         library primitive; the host cuts the runs at 255 positions (:267) and returns the intervals.
#include <hipcub/hipcub.hpp>

#include "kvm_internal.h"
#include "kvm_seqadd.h"

#define IDX_EPOCH 100000   // K/IndexBuilder.java:136
#define IDX_THREADS 256
#define IDX_STEPS 1024     // steps per LDS chunk: D[IDX_STEPS] | O[IDX_STEPS], two buffers = 32 KiB

// steps of one EPOCH buffer: step i reads the buffer's point i (0-based global index base + i); for i >= w - 1 it
// completes the window at position base + i - w + 1 (0-based), and the point leaving afterwards is base + i - w + 1.
__global__ __launch_bounds__(IDX_THREADS) void kvm_index_ex_kernel(const double *__restrict__ x, int64_t n, int32_t w,
                                                                   double *__restrict__ ex_out) {
    __shared__ __attribute__((aligned(16))) double bufs[2][2 * IDX_STEPS];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * (IDX_EPOCH - w + 1);
    const int64_t ep = min((int64_t)IDX_EPOCH, n - base);  // points in this buffer
    if (ep <= w - 1) return;                                // K/IndexBuilder.java:224
    const bool consumer = tid < 64;
    const int ptid = tid - 64, pthreads = IDX_THREADS - 64;
    auto produce = [&](int64_t c0, int b) {
        constexpr int PER = (IDX_STEPS + (IDX_THREADS - 64) - 1) / (IDX_THREADS - 64);
        double d[PER], o[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {  // loads first (unconditional, clamped), stores after
            const int64_t s = c0 + ptid + i * pthreads;
            const int64_t sd = s < ep ? s : ep - 1, so = s >= w - 1 ? s - (w - 1) : 0;
            const double dv = x[base + sd], ov = x[base + (so < ep ? so : ep - 1)];
            d[i] = s < ep ? dv : 0.0;
            o[i] = (s < ep && s >= w - 1) ? ov : 0.0;
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int t = ptid + i * pthreads;
            if (t < IDX_STEPS) {
                bufs[b][t] = d[i];
                bufs[b][IDX_STEPS + t] = o[i];
            }
        }
    };
    const int nch = (int)((ep + IDX_STEPS - 1) / IDX_STEPS);
    double ex = 0.0;  // K/IndexBuilder.java:232
    if (!consumer) produce(0, 0);
    __syncthreads();
    for (int c = 0; c < nch; c++) {
        const int64_t c0 = (int64_t)c * IDX_STEPS;
        if (consumer) {
            if (tid == 0) {
                const double *dbuf = bufs[c & 1], *obuf = dbuf + IDX_STEPS;
                const int cnt = (int)min((int64_t)IDX_STEPS, ep - c0);
                const int n_add = (int)max((int64_t)0, min((int64_t)cnt, (int64_t)(w - 1) - c0));  // no window complete yet
                ex = ex_seq_add(dbuf, n_add, ex);                                                   // :239
                // window steps, branch-free, LDS reads one round ahead (see kvm_exact.hip)
                constexpr int R = 4;
                double dv[2][R], ov[2][R];
                double a_l = ex, dummy = 0.0;
                double *out = ex_out + (base + c0 - (int64_t)(w - 1));  // + t = position completed by step c0 + t
                const int nw = cnt - n_add, rounds = nw / (2 * R);
                auto fetch = [&](int t0, int set) {
#pragma unroll
                    for (int u = 0; u < R; u++) {
                        dv[set][u] = dbuf[t0 + u];
                        ov[set][u] = obuf[t0 + u];
                    }
                };
                auto consume = [&](int t0, int set) {
#pragma unroll
                    for (int u = 0; u < R; u++) {
                        a_l += dv[set][u];      // ex += d           K/IndexBuilder.java:239
                        out[t0 + u] = a_l;      // -> mean = ex / w  :250 (formed by kvm_index_keys_kernel)
                        a_l -= ov[set][u];      // ex -= t[j]        :287
                    }
                };
                if (rounds > 0) fetch(n_add, 0);
                for (int rr = 0; rr < rounds; rr++) {
                    const int t = n_add + rr * 2 * R;
                    fetch(t + R, 1);
                    asm volatile("" : "+v"(a_l), "+v"(dummy)::"memory");
                    consume(t, 0);
                    fetch((rr + 1 < rounds) ? t + 2 * R : n_add, 0);
                    asm volatile("" : "+v"(a_l), "+v"(dummy)::"memory");
                    consume(t + R, 1);
                }
                for (int t = n_add + rounds * 2 * R; t < cnt; t++) {
                    a_l += dbuf[t];
                    out[t] = a_l;
                    a_l -= obuf[t];
                }
                ex = a_l;
            }
        } else if (c + 1 < nch) {
            produce(c0 + IDX_STEPS, (c + 1) & 1);
        }
        __syncthreads();
    }
}

// MeanIntervalUtils.toRound, K/utils/MeanIntervalUtils.java:50-60 (posOfD = 2: Math.pow(10, 1) = 10.0, Math.pow(10, -1) = 0.1)
__device__ __forceinline__ double kvm_to_round(double value) {
    value *= 10.0;
    const double intValue = floor(value);
    const double diff = value - intValue;
    double ret = intValue;
    if (diff >= 0.5 || diff != diff) ret += 0.5;  // Double.compare(diff, 0.5) >= 0 (NaN compares greater)
    ret *= 0.1;
    return ret;
}

__global__ __launch_bounds__(256) void kvm_index_keys_kernel(const double *__restrict__ ex, int64_t count, int32_t w,
                                                             unsigned long long *__restrict__ key_bits) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const double mean = ex[i] / w;  // K/IndexBuilder.java:250
        double key = kvm_to_round(mean);
        unsigned long long u = (unsigned long long)__double_as_longlong(key);
        if (key != key) u = 0x7ff8000000000000ULL;  // Double.equals: every NaN is the same key
        key_bits[i] = u;
    }
}

// pieces of one run of equal keys: an interval holds at most MAXIMUM_DIFF - 1 = 255 positions (K/IndexBuilder.java:267)
__global__ __launch_bounds__(256) void kvm_index_pieces_kernel(const int32_t *__restrict__ len, int32_t num, int32_t *__restrict__ pieces) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < num) pieces[i] = (len[i] + 254) / 255;
}

__global__ __launch_bounds__(256) void kvm_index_expand_kernel(const unsigned long long *__restrict__ unique, const int32_t *__restrict__ len,
                                                               const int32_t *__restrict__ pos0, const int32_t *__restrict__ off,
                                                               int32_t num, unsigned long long *__restrict__ keys,
                                                               int32_t *__restrict__ first, int32_t *__restrict__ last) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num) return;
    const unsigned long long k = unique[i];
    const int32_t p = pos0[i] + 1, end = pos0[i] + len[i];  // 1-based window positions of the run
    int32_t o = off[i];
    for (int32_t f = p; f <= end; f += 255, o++) {
        keys[o] = k;
        first[o] = f;
        last[o] = min(f + 254, end);
    }
}

void kvm_index_dev_free(kvm_index_dev *d) {
    for (void *&b : d->blocks) {
        if (b) hipFree(b);
        b = nullptr;
    }
    d->keys = d->group_keys = nullptr;
    d->first = d->last = d->group_off = nullptr;
    d->total = d->n_groups = 0;
}

// The whole device pass: running sums -> keys -> runs of equal keys -> intervals of <= 255 positions, left in device
// memory (d->keys / first / last, d->total of them) until kvm_index_dev_free.
int kvm_index_intervals_device(kvm_ctx *ctx, int32_t w, int64_t count, kvm_index_dev *d, hipStream_t s) {
    *d = kvm_index_dev{};
    unsigned long long *d_keys = nullptr, *d_unique = nullptr;
    int32_t *d_len = nullptr, *d_small = nullptr;
    auto alloc = [&](void **p, size_t bytes, int slot) {
        if (hipMalloc(p, bytes + 64) != hipSuccess) return false;
        d->blocks[slot] = *p;
        return true;
    };
    if (!alloc((void **)&d_keys, (size_t)count * 8, 0) || !alloc((void **)&d_unique, (size_t)count * 8, 1) ||
        !alloc((void **)&d_len, (size_t)count * 4, 2) || !alloc((void **)&d_small, 64, 3)) {
        kvm_index_dev_free(d);
        return KVM_ERR_DEVICE;
    }
    int32_t *d_num = d_small;
    // 1. exact running sums per EPOCH buffer -> ex per position (into d_unique's storage as scratch: same size)
    double *d_ex = reinterpret_cast<double *>(d_unique);
    const int64_t per = IDX_EPOCH - w + 1;
    const int64_t buffers = (count + per - 1) / per;
    hipLaunchKernelGGL(kvm_index_ex_kernel, dim3((unsigned)buffers), dim3(IDX_THREADS), 0, s, ctx->d_series, ctx->n, w, d_ex);
    // 2. keys
    const int64_t kb = std::min<int64_t>((count + 255) / 256, 65536);
    hipLaunchKernelGGL(kvm_index_keys_kernel, dim3((unsigned)kb), dim3(256), 0, s, d_ex, count, w, d_keys);
    bool ok = hipGetLastError() == hipSuccess;
    // 3. run-length encode (Double.equals on the bit patterns)
    size_t temp_bytes = 0, tb2 = 0;
    void *d_temp = nullptr;
    ok = ok && hipcub::DeviceRunLengthEncode::Encode(nullptr, temp_bytes, d_keys, d_unique, d_len, d_num, (int)count, s) == hipSuccess;
    ok = ok && hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, d_len, d_len, (int)count, s) == hipSuccess;
    temp_bytes = std::max(temp_bytes, tb2);
    ok = ok && alloc(&d_temp, temp_bytes, 4);
    ok = ok && hipcub::DeviceRunLengthEncode::Encode(d_temp, temp_bytes, d_keys, d_unique, d_len, d_num, (int)count, s) == hipSuccess;
    int32_t num = 0;
    ok = ok && hipMemcpyAsync(&num, d_num, 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    if (!ok || num <= 0 || num > count) {
        kvm_index_dev_free(d);
        return KVM_ERR_DEVICE;
    }
    // 4. cut the runs at 255 positions: piece counts, the two exclusive scans, expansion
    int32_t *d_pieces = nullptr, *d_pos0 = nullptr, *d_off = nullptr;
    ok = alloc((void **)&d_pieces, (size_t)num * 4, 5) && alloc((void **)&d_pos0, (size_t)num * 4, 6) && alloc((void **)&d_off, (size_t)num * 4, 7);
    if (ok) {
        hipLaunchKernelGGL(kvm_index_pieces_kernel, dim3((unsigned)((num + 255) / 256)), dim3(256), 0, s, d_len, num, d_pieces);
        size_t tb = temp_bytes;
        ok = hipcub::DeviceScan::ExclusiveSum(d_temp, tb, d_len, d_pos0, num, s) == hipSuccess;
        tb = temp_bytes;
        ok = ok && hipcub::DeviceScan::ExclusiveSum(d_temp, tb, d_pieces, d_off, num, s) == hipSuccess;
    }
    int32_t tail[2] = {0, 0};
    ok = ok && hipMemcpyAsync(&tail[0], d_off + (num - 1), 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipMemcpyAsync(&tail[1], d_pieces + (num - 1), 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    const int64_t total = (int64_t)tail[0] + tail[1];
    if (!ok || total < num || total > count) {
        kvm_index_dev_free(d);
        return KVM_ERR_DEVICE;
    }
    // outputs: the keys go where the per-position keys were (total <= count), first / last get their own block
    int32_t *d_fl = nullptr;
    if (!alloc((void **)&d_fl, (size_t)total * 8, 8)) {
        kvm_index_dev_free(d);
        return KVM_ERR_DEVICE;
    }
    hipLaunchKernelGGL(kvm_index_expand_kernel, dim3((unsigned)((num + 255) / 256)), dim3(256), 0, s, d_unique, d_len, d_pos0, d_off, num,
                       d_keys, d_fl, d_fl + total);
    if (hipGetLastError() != hipSuccess) {
        kvm_index_dev_free(d);
        return KVM_ERR_DEVICE;
    }
    d->keys = d_keys;
    d->first = d_fl;
    d->last = d_fl + total;
    d->total = total;
    return KVM_OK;
}

// ---- grouping: the rows of step 1 (K/IndexBuilder.java:270-284 collects the intervals of one key in one IndexNode; step 2,
// :310-318, walks the rows by DESCENDING key) -------------------------------------------------------------------------
// Sort key: the bit pattern mapped so that unsigned order == Double.compareTo order (-0.0 < 0.0, NaN above +Inf),
// complemented for descending order.  The radix sort is stable, so the intervals of a key stay in position order.
__device__ __forceinline__ unsigned long long kvm_key_to_sortable_desc(unsigned long long u) {
    const unsigned long long asc = (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
    return ~asc;
}
__device__ __forceinline__ unsigned long long kvm_sortable_desc_to_key(unsigned long long v) {
    const unsigned long long asc = ~v;
    return (asc >> 63) ? (asc & 0x7fffffffffffffffULL) : ~asc;
}
__global__ __launch_bounds__(256) void kvm_index_pack_kernel(const unsigned long long *__restrict__ keys, const int32_t *__restrict__ first,
                                                             const int32_t *__restrict__ last, int64_t total,
                                                             unsigned long long *__restrict__ skey, unsigned long long *__restrict__ val) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        skey[i] = kvm_key_to_sortable_desc(keys[i]);
        val[i] = ((unsigned long long)(uint32_t)first[i] << 32) | (uint32_t)last[i];
    }
}
__global__ __launch_bounds__(256) void kvm_index_unpack_kernel(const unsigned long long *__restrict__ skey, const unsigned long long *__restrict__ val,
                                                               int64_t total, unsigned long long *__restrict__ keys,
                                                               int32_t *__restrict__ first, int32_t *__restrict__ last) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = kvm_sortable_desc_to_key(skey[i]);
        first[i] = (int32_t)(val[i] >> 32);
        last[i] = (int32_t)(val[i] & 0xffffffffULL);
    }
}
__global__ __launch_bounds__(256) void kvm_index_group_keys_kernel(unsigned long long *__restrict__ gk, int32_t n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gk[i] = kvm_sortable_desc_to_key(gk[i]);
}

int kvm_index_group_device(kvm_index_dev *d, hipStream_t s) {
    const int64_t total = d->total;
    if (total <= 0) return KVM_OK;
    int slot = 9;
    auto alloc = [&](void **p, size_t bytes) {
        if (slot >= 16 || hipMalloc(p, bytes + 64) != hipSuccess) return false;
        d->blocks[slot++] = *p;
        return true;
    };
    unsigned long long *sk_in = nullptr, *sk_out = nullptr, *v_in = nullptr, *v_out = nullptr;
    int32_t *g_cnt = nullptr, *g_num = nullptr;
    bool ok = alloc((void **)&sk_in, (size_t)total * 8) && alloc((void **)&sk_out, (size_t)total * 8) &&
              alloc((void **)&v_in, (size_t)total * 8) && alloc((void **)&v_out, (size_t)total * 8) &&
              alloc((void **)&g_cnt, (size_t)(total + 1) * 4) && alloc((void **)&g_num, 64);
    if (!ok) return KVM_ERR_DEVICE;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(kvm_index_pack_kernel, dim3(grid), dim3(256), 0, s, d->keys, d->first, d->last, total, sk_in, v_in);
    size_t tb_sort = 0, tb_rle = 0, tb_scan = 0;
    ok = hipcub::DeviceRadixSort::SortPairs(nullptr, tb_sort, sk_in, sk_out, v_in, v_out, (int)total, 0, 64, s) == hipSuccess &&
         hipcub::DeviceRunLengthEncode::Encode(nullptr, tb_rle, sk_out, sk_in, g_cnt, g_num, (int)total, s) == hipSuccess &&
         hipcub::DeviceScan::ExclusiveSum(nullptr, tb_scan, g_cnt, g_cnt, (int)total + 1, s) == hipSuccess;
    size_t tb = std::max(tb_sort, std::max(tb_rle, tb_scan));
    void *d_temp = nullptr;
    ok = ok && alloc(&d_temp, tb);
    size_t t1 = tb, t2 = tb, t3 = tb;
    ok = ok && hipcub::DeviceRadixSort::SortPairs(d_temp, t1, sk_in, sk_out, v_in, v_out, (int)total, 0, 64, s) == hipSuccess;
    if (ok) hipLaunchKernelGGL(kvm_index_unpack_kernel, dim3(grid), dim3(256), 0, s, sk_out, v_out, total, d->keys, d->first, d->last);
    // distinct keys and their interval counts (sk_in and v_in are free again: group keys / offsets go there)
    ok = ok && hipcub::DeviceRunLengthEncode::Encode(d_temp, t2, sk_out, sk_in, g_cnt, g_num, (int)total, s) == hipSuccess;
    int32_t ng = 0;
    ok = ok && hipMemcpyAsync(&ng, g_num, 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    if (!ok || ng <= 0 || ng > total) return KVM_ERR_DEVICE;
    int32_t *g_off = reinterpret_cast<int32_t *>(v_in);
    ok = hipMemsetAsync(g_cnt + ng, 0, 4, s) == hipSuccess &&
         hipcub::DeviceScan::ExclusiveSum(d_temp, t3, g_cnt, g_off, ng + 1, s) == hipSuccess;
    hipLaunchKernelGGL(kvm_index_group_keys_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, sk_in, ng);
    if (!ok || hipGetLastError() != hipSuccess) return KVM_ERR_DEVICE;
    d->group_keys = sk_in;
    d->group_off = g_off;
    d->n_groups = ng;
    return KVM_OK;
}
