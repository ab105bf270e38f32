// kvm_api_index.hip -- the C ABI of the index build, step 1 (K/IndexBuilder.java:186-300): kvm_index_rows_* over kvm_index.hip
#include "kvm_api_internal.h"

// ---- index build, step 1 (kvm_index.hip) ----------------------------------------------------------------------
// The result lives in one pinned host block (keys | first | last) that the device-to-host copies land in directly; one
// freed block is kept per process for the next build (the five window lengths of one index build reuse it).
struct kvm_index_rows {
    void *block = nullptr;
    size_t cap = 0;
    int64_t count = 0;
    const double *keys = nullptr;
    const int32_t *first = nullptr, *last = nullptr;
    int64_t n_groups = 0;               // grouped build only
    const double *group_keys = nullptr;
    const int32_t *group_off = nullptr;
};
static std::mutex g_idx_mu;
static void *g_idx_block = nullptr;
static size_t g_idx_cap = 0;

static void *idx_block_take(size_t bytes, size_t *cap) {
    {
        std::lock_guard<std::mutex> lk(g_idx_mu);
        if (g_idx_block && g_idx_cap >= bytes) {
            void *p = g_idx_block;
            *cap = g_idx_cap;
            g_idx_block = nullptr;
            g_idx_cap = 0;
            return p;
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
    *cap = bytes;
    return p;
}
static void idx_block_give(void *p, size_t cap) {
    if (!p) return;
    void *drop = p;
    {
        std::lock_guard<std::mutex> lk(g_idx_mu);
        if (cap > g_idx_cap) {
            drop = g_idx_block;
            g_idx_block = p;
            g_idx_cap = cap;
        }
    }
    if (drop) hipHostFree(drop);
}
void kvm_idx_block_drop() {
    void *p = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_idx_mu);
        p = g_idx_block;
        g_idx_block = nullptr;
        g_idx_cap = 0;
    }
    if (p) hipHostFree(p);
}

static int index_rows_build(kvm_ctx *ctx, int32_t w, bool grouped, kvm_index_rows **out, bool with_keys = true) {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!out) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "null result pointer");
    *out = nullptr;
    if (!ctx->d_series) return kvm_fail(ctx, KVM_ERR_NO_SERIES, "no series loaded");
    if (w < 1 || w > 100000 || (int64_t)w > ctx->n)
        return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_index_rows_build: window length outside [1, min(n, EPOCH)]");
    const int64_t count = ctx->n - w + 1;
    if (count > 0x7fffffffLL) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_index_rows_build: series too long");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    kvm_index_dev dev;
    if (ctx->opt_timing) hipEventRecord(ctx->ev0, ctx->stream);
    int rc = kvm_index_intervals_device(ctx, w, count, &dev, ctx->stream);
    if (rc) return kvm_fail(ctx, rc, "kvm_index_rows_build: device pass failed (out of device memory?)");
    if (grouped && (rc = kvm_index_group_device(&dev, ctx->stream))) {
        kvm_index_dev_free(&dev);
        return kvm_fail(ctx, rc, "kvm_index_rows_build_grouped: grouping pass failed (out of device memory?)");
    }
    if (ctx->opt_timing) hipEventRecord(ctx->ev1, ctx->stream);
    kvm_index_rows *r = new kvm_index_rows();
    const size_t total = (size_t)dev.total, keys_bytes = (total * 8 + 63) & ~(size_t)63, fl_bytes = (total * 4 + 63) & ~(size_t)63;
    const size_t ng = (size_t)dev.n_groups, gk_bytes = (ng * 8 + 63) & ~(size_t)63, go_bytes = ((ng + 1) * 4 + 63) & ~(size_t)63;
    r->block = idx_block_take(keys_bytes + 2 * fl_bytes + gk_bytes + go_bytes + 64, &r->cap);
    if (!r->block) {
        kvm_index_dev_free(&dev);
        delete r;
        return kvm_fail(ctx, KVM_ERR_DEVICE, "kvm_index_rows_build: out of pinned host memory");
    }
    char *base = (char *)r->block;
    char *gbase = base + keys_bytes + 2 * fl_bytes;
    // (the per-interval keys are half of the bytes; the file builder reads the group keys only and leaves them on the device)
    const bool ok = (!with_keys || hipMemcpyAsync(base, dev.keys, total * 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess) &&
                    hipMemcpyAsync(base + keys_bytes, dev.first, total * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                    hipMemcpyAsync(base + keys_bytes + fl_bytes, dev.last, total * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                    (ng == 0 || (hipMemcpyAsync(gbase, dev.group_keys, ng * 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                                 hipMemcpyAsync(gbase + gk_bytes, dev.group_off, (ng + 1) * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess)) &&
                    hipStreamSynchronize(ctx->stream) == hipSuccess;
    kvm_index_dev_free(&dev);
    if (!ok) {
        idx_block_give(r->block, r->cap);
        delete r;
        return kvm_fail(ctx, KVM_ERR_DEVICE, "kvm_index_rows_build: readback failed");
    }
    if (ctx->opt_timing) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_kernel_ns = (int64_t)(ms * 1e6);
    }
    r->count = (int64_t)total;
    r->keys = with_keys ? (const double *)base : nullptr;
    r->first = (const int32_t *)(base + keys_bytes);
    r->last = (const int32_t *)(base + keys_bytes + fl_bytes);
    if (ng) {
        r->n_groups = (int64_t)ng;
        r->group_keys = (const double *)gbase;
        r->group_off = (const int32_t *)(gbase + gk_bytes);
    }
    *out = r;
    return KVM_OK;
}
extern "C" int kvm_index_rows_build(kvm_ctx *ctx, int32_t w, kvm_index_rows **out) { return kvm_guard<int>(ctx, [&]() -> int { return index_rows_build(ctx, w, false, out); }); }
extern "C" int kvm_index_rows_build_grouped(kvm_ctx *ctx, int32_t w, kvm_index_rows **out) { return kvm_guard<int>(ctx, [&]() -> int { return index_rows_build(ctx, w, true, out); }); }
// kvm_index_build's own call (kvm_index_file.cpp; C++ linkage, not part of the ABI): grouped rows without the per-interval keys
int kvm_index_rows_build_for_file(kvm_ctx *ctx, int32_t w, kvm_index_rows **out) { return kvm_guard<int>(ctx, [&]() -> int { return index_rows_build(ctx, w, true, out, false); }); }
extern "C" int64_t kvm_index_rows_groups(const kvm_index_rows *r) { return r ? r->n_groups : 0; }
extern "C" const double *kvm_index_rows_group_keys(const kvm_index_rows *r) { return r ? r->group_keys : nullptr; }
extern "C" const int32_t *kvm_index_rows_group_offsets(const kvm_index_rows *r) { return r ? r->group_off : nullptr; }
extern "C" int64_t kvm_index_rows_count(const kvm_index_rows *r) { return r ? r->count : 0; }
extern "C" const double *kvm_index_rows_keys(const kvm_index_rows *r) { return r ? r->keys : nullptr; }
extern "C" const int32_t *kvm_index_rows_first(const kvm_index_rows *r) { return r ? r->first : nullptr; }
extern "C" const int32_t *kvm_index_rows_last(const kvm_index_rows *r) { return r ? r->last : nullptr; }
extern "C" void kvm_index_rows_free(kvm_index_rows *r) {
    if (!r) return;
    idx_block_give(r->block, r->cap);
    delete r;
}

