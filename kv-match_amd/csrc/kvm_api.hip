// kvm_api.hip -- the C ABI of include/kvmatch.h: context, series staging, the four phase-2 entry
// points and result objects.  Host orchestration only; every distance is computed by the HIP kernels in
// kvm_ed*.hip / kvm_dtw*.hip.  There is deliberately no CPU fallback: every entry point fails with
// KVM_ERR_DEVICE when HIP is unusable.
#include "kvm_guard.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <numeric>

#include "kvm_internal.h"
#include "kvm_jsort.h"

#define KVM_MAX_SCAN_DATA_LENGTH 40000  // K/NormQueryEngine.java:60, K/NormQueryEngineDtw.java:61
#define KVM_TILE_MAX 64                 // candidates per tile: wave ED kernel, DTW filter (one lane each)
#define KVM_TILE_TILED 32               // candidates per tile of the LDS-tiled ED kernel (== ED_ROW_CAND)

static thread_local std::string g_open_err;

static void idx_block_drop();  // pinned result block of the index build, kept per process

static int fail(kvm_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg; else g_open_err = msg;
    return code;
}

void kvm_set_error(kvm_ctx *ctx, const std::string &msg) { fail(ctx, 0, msg); }  // for the host-only translation units
void kvm_set_error_noexcept(kvm_ctx *ctx, const char *msg) noexcept {
    try {
        (ctx ? ctx->err : g_open_err).assign(msg);
    } catch (...) {  // (not even the message fits: the status code has to do)
    }
}

#define HIPCHK(ctx, call)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(ctx, KVM_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)

static int ensure(kvm_ctx *ctx, kvm_dev_buf &b, size_t bytes) {
    if (bytes <= b.cap) return KVM_OK;
    if (b.p) HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return KVM_OK;
}

static int ensure_pinned(kvm_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->h_pinned_cap) return KVM_OK;
    if (ctx->h_pinned) HIPCHK(ctx, hipHostFree(ctx->h_pinned));
    ctx->h_pinned = nullptr;
    ctx->h_pinned_cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(ctx, hipHostMalloc(&ctx->h_pinned, want, hipHostMallocDefault));
    ctx->h_pinned_cap = want;
    return KVM_OK;
}

extern "C" const char *kvm_version(void) { return "kvmatch-hip 0.1 (gfx950)"; }

extern "C" const char *kvm_last_error(const kvm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_open_err.c_str(); }

extern "C" int kvm_open(int device, kvm_ctx **out) {
    return kvm_guard<int>(nullptr, [&]() -> int {
    if (!out) return fail(nullptr, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_open: null out pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, KVM_ERR_DEVICE,
                    std::string("kvm_open: no usable HIP device (") + hipGetErrorString(e) +
                        "); this library has no CPU fallback");
    if (device < 0 || device >= ndev)
        return fail(nullptr, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_open: device ordinal out of range");
    kvm_ctx *ctx = new kvm_ctx();
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&ctx->stream)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_staged, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming)) != hipSuccess) {
        std::string msg = std::string("kvm_open: ") + hipGetErrorString(e);
        if (ctx->ev0) hipEventDestroy(ctx->ev0);
        if (ctx->ev1) hipEventDestroy(ctx->ev1);
        if (ctx->ev_staged) hipEventDestroy(ctx->ev_staged);
        if (ctx->ev_done) hipEventDestroy(ctx->ev_done);
        if (ctx->stream) hipStreamDestroy(ctx->stream);
        delete ctx;
        return fail(nullptr, KVM_ERR_DEVICE, msg);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (std::strstr(prop.gcnArchName, "gfx950") == nullptr) {
            std::string msg = std::string("kvm_open: built for gfx950 only, device is ") + prop.gcnArchName;
            hipEventDestroy(ctx->ev0);
            hipEventDestroy(ctx->ev1);
            hipEventDestroy(ctx->ev_staged);
    hipEventDestroy(ctx->ev_done);
            hipEventDestroy(ctx->ev_done);
            hipStreamDestroy(ctx->stream);
            delete ctx;
            return fail(nullptr, KVM_ERR_DEVICE, msg);
        }
    }
    *out = ctx;
    return KVM_OK;
    });
}

static void free_series(kvm_ctx *ctx) {
    if (ctx->own_series && ctx->d_series) hipFree(ctx->d_series);
    ctx->d_series = nullptr;
    if (ctx->own_series && ctx->d_prefix) hipFree(ctx->d_prefix);  // (a shared series belongs to its owner context)
    if (ctx->own_series && ctx->d_blockmax) hipFree(ctx->d_blockmax);
    ctx->own_series = false;
    ctx->d_prefix = nullptr;
    ctx->d_blockmax = nullptr;
    if (ctx->d_env_u) hipFree(ctx->d_env_u);
    if (ctx->d_env_l) hipFree(ctx->d_env_l);
    ctx->d_env_u = ctx->d_env_l = nullptr;
    ctx->env_rho = -1;
    ctx->n = 0;
}

extern "C" int kvm_close(kvm_ctx *ctx) {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    free_series(ctx);
    kvm_dev_buf *bufs[] = {&ctx->b_query, &ctx->b_tiles, &ctx->b_out_off, &ctx->b_out_dist, &ctx->b_counters,
                           &ctx->b_cands, &ctx->b_cands2, &ctx->b_aux,   &ctx->b_aux2,    &ctx->b_scratch, &ctx->b_plans,
                           &ctx->b_ucr};
    for (kvm_dev_buf *b : bufs)
        if (b->p) hipFree(b->p);
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    if (ctx->h_readback) hipHostFree(ctx->h_readback);
    if (ctx->h_readback2) hipHostFree(ctx->h_readback2);
    idx_block_drop();
    hipEventDestroy(ctx->ev0);
    hipEventDestroy(ctx->ev1);
    hipEventDestroy(ctx->ev_staged);
    if (ctx->ev_env) hipEventDestroy(ctx->ev_env);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return KVM_OK;
}

extern "C" int kvm_set_option(kvm_ctx *ctx, int32_t option, int64_t value) {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    switch (option) {
        case KVM_OPT_EARLY_ABANDON: ctx->opt_early_abandon = value ? 1 : 0; return KVM_OK;
        case KVM_OPT_LB_CASCADE: ctx->opt_lb_cascade = value ? 1 : 0; return KVM_OK;
        case KVM_OPT_ED_KERNEL:
            if (value != 0 && value != 1 && value != 3)
                return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "KVM_OPT_ED_KERNEL in {0,1,3}");
            ctx->opt_ed_kernel = value;
            return KVM_OK;
        case KVM_OPT_TIMING: ctx->opt_timing = value ? 1 : 0; return KVM_OK;
        case KVM_OPT_DP_KERNEL: ctx->opt_dp_kernel = (value >= 0 && value <= 3) ? value : 0; return KVM_OK;
        case KVM_OPT_SORT_VIOLATION: ctx->opt_sort_violation = value ? 1 : 0; return KVM_OK;
        case KVM_OPT_UCR_LOOPS: ctx->opt_ucr_loops = value ? 1 : 0; return KVM_OK;
        case KVM_OPT_STRICT_BOUNDS: ctx->opt_strict_bounds = value ? 1 : 0; return KVM_OK;
    }
    return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "unknown option");
}

// The events around the timed kernel are read only once everything behind them has been enqueued (and, on the host path,
// after the call's single synchronisation): waiting for them inside the pipeline would put a host wake-up between the
// kernels of a call.
static void finish_timing(kvm_ctx *ctx) {
    if (!ctx->timing_pending) return;
    ctx->timing_pending = false;
    float ms = 0;
    if (hipEventSynchronize(ctx->ev1) == hipSuccess && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess)
        ctx->last_kernel_ns = (int64_t)((double)ms * 1e6);
}
extern "C" int kvm_measure_device(kvm_ctx *ctx, int32_t what, double *value) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!value || what < 0 || what > 2) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_measure_device: what = 0, 1 or 2 and a result pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, kvm_measure_device_rate(what, ctx->num_cu, ctx->stream, ctx->ev0, ctx->ev1, value));
    return KVM_OK;
    });
}

int64_t kvm_last_kernel_ns(const kvm_ctx *ctx) {
    if (!ctx) return 0;
    finish_timing(const_cast<kvm_ctx *>(ctx));
    return ctx->last_kernel_ns;
}

extern "C" int64_t kvm_get_option(const kvm_ctx *ctx, int32_t option) {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    switch (option) {
        case KVM_OPT_EARLY_ABANDON: return ctx->opt_early_abandon;
        case KVM_OPT_LB_CASCADE: return ctx->opt_lb_cascade;
        case KVM_OPT_ED_KERNEL: return ctx->opt_ed_kernel;
        case KVM_OPT_TIMING: return ctx->opt_timing;
        case KVM_OPT_DP_KERNEL: return ctx->opt_dp_kernel;
        case KVM_OPT_SORT_VIOLATION: return ctx->opt_sort_violation;
        case KVM_OPT_UCR_LOOPS: return ctx->opt_ucr_loops;
        case KVM_OPT_STRICT_BOUNDS: return ctx->opt_strict_bounds;
    }
    return KVM_ERR_ILLEGAL_ARGUMENT;
}

// ---- series staging -----------------------------------------------------------------------------------

static int finish_load_steps(kvm_ctx *ctx) {
    // prefix sums for per-window mean/std (replaces the running ex/ex2 of K/NormQueryEngine.java:497-509)
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_prefix, (size_t)ctx->n * sizeof(double2)));
    HIPCHK(ctx, kvm_launch_prefix(ctx->d_series, ctx->d_prefix, ctx->n, ctx->stream));
    // largest x^2 per prefix block: what the reference's running sums may still carry the residue of (kvm_hist_*)
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_blockmax, (size_t)((ctx->n >> KVM_PS_BLOCK_LOG2) + 1) * sizeof(double)));
    HIPCHK(ctx, kvm_launch_blockmax(ctx->d_series, ctx->n, ctx->d_blockmax, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return KVM_OK;
}
// A series is either fully staged (values + prefix sums) or absent: on any failure (16 B per point of prefix sums may not
// fit) the context drops the half-loaded series, so that later calls report KVM_ERR_NO_SERIES instead of reading a null
// prefix array.
static int finish_load(kvm_ctx *ctx) {
    const int rc = finish_load_steps(ctx);
    if (rc != KVM_OK) {
        hipStreamSynchronize(ctx->stream);
        free_series(ctx);
    }
    return rc;
}

static int alloc_series(kvm_ctx *ctx, int64_t n) {
    if (n < 1 || n > 2147483647LL)  // offsets are Java ints (K/operator/file/TimeSeriesFileOperator.java:32)
        return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "series length must be in [1, 2^31-1]");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    free_series(ctx);
    // KVM_SERIES_PAD zero points past n let the tile kernels stage whole rows without bounds checks
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_series, (size_t)(n + KVM_SERIES_PAD) * sizeof(double)));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_series + n, 0, (size_t)KVM_SERIES_PAD * sizeof(double), ctx->stream));
    ctx->own_series = true;
    ctx->n = n;
    return KVM_OK;
}

extern "C" int kvm_load_series_f64(kvm_ctx *ctx, const double *host_values, int64_t n) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!host_values) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "null series");
    int rc = alloc_series(ctx, n);
    if (rc) return rc;
    if (hipMemcpyAsync(ctx->d_series, host_values, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        free_series(ctx);
        return fail(ctx, KVM_ERR_DEVICE, "copying the series to the device failed");
    }
    return finish_load(ctx);
    });
}

extern "C" int kvm_load_series_be64(kvm_ctx *ctx, const void *be_bytes, int64_t n) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!be_bytes) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "null series");
    int rc = alloc_series(ctx, n);
    if (rc) return rc;
    if (hipMemcpyAsync(ctx->d_series, be_bytes, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        kvm_launch_bswap64((uint64_t *)ctx->d_series, n, ctx->stream) != hipSuccess) {
        hipStreamSynchronize(ctx->stream);
        free_series(ctx);
        return fail(ctx, KVM_ERR_DEVICE, "staging the big-endian series failed");
    }
    return finish_load(ctx);
    });
}

extern "C" int kvm_load_series_file(kvm_ctx *ctx, const char *path, int64_t n) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!path) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "null path");
    if (n < 1 || n > 2147483647LL)  // (before the file is touched: a bad length is the caller's error, not an I/O error)
        return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "series length must be in [1, 2^31-1]");
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(ctx, KVM_ERR_IO, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || (int64_t)st.st_size < n * 8) {
        close(fd);
        return fail(ctx, KVM_ERR_IO, std::string(path) + ": shorter than n*8 bytes");
    }
    void *map = mmap(nullptr, (size_t)n * 8, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return fail(ctx, KVM_ERR_IO, std::string("mmap failed for ") + path);
    int rc = kvm_load_series_be64(ctx, map, n);
    munmap(map, (size_t)n * 8);
    return rc;
    });
}

extern "C" int kvm_attach_series_device(kvm_ctx *ctx, const double *device_values, int64_t n) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!device_values || n < 1 || n > 2147483647LL) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "bad device series");
    // device-to-device copy into the ctx's own padded buffer (see alloc_series); the caller's array is not
    // referenced after this call returns
    int rc = alloc_series(ctx, n);
    if (rc) return rc;
    if (hipMemcpyAsync(ctx->d_series, device_values, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
        free_series(ctx);
        return fail(ctx, KVM_ERR_DEVICE, "copying the device series failed");
    }
    return finish_load(ctx);
    });
}

extern "C" int kvm_share_series(kvm_ctx *ctx, const kvm_ctx *owner) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx || !owner) return KVM_ERR_INVALID_HANDLE;
    if (ctx == owner) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "a context cannot share its own series");
    if (!owner->d_series || !owner->d_prefix) return fail(ctx, KVM_ERR_NO_SERIES, "the owner context has no series loaded");
    if (owner->device != ctx->device) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "contexts on different devices cannot share a series");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    free_series(ctx);
    ctx->d_series = owner->d_series;  // values and prefix sums are read-only after staging
    ctx->d_prefix = owner->d_prefix;
    ctx->d_blockmax = owner->d_blockmax;
    ctx->own_series = false;
    ctx->n = owner->n;
    return KVM_OK;
    });
}

extern "C" int64_t kvm_series_length(const kvm_ctx *ctx) { return ctx ? ctx->n : KVM_ERR_INVALID_HANDLE; }

extern "C" int kvm_read_time_series(kvm_ctx *ctx, int32_t left, int32_t length, double *out) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!ctx->d_series) return fail(ctx, KVM_ERR_NO_SERIES, "no series loaded");
    // K/operator/file/TimeSeriesFileOperator.java:55-57
    if (left < 1 || (int64_t)left + length - 1 > ctx->n || length < 1 || !out) {
        char msg[96];
        snprintf(msg, sizeof msg, "left: %d, length: %d", left, length);
        return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, msg);
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_series + (left - 1), (size_t)length * 8, hipMemcpyDeviceToHost,
                               ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return KVM_OK;
    });
}

extern "C" int kvm_window_means(kvm_ctx *ctx, int32_t w, int32_t first, int64_t count, double *out) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!ctx->d_series) return fail(ctx, KVM_ERR_NO_SERIES, "no series loaded");
    if (w < 1 || first < 1 || count < 0 || !out || (int64_t)first + count - 1 + w - 1 > ctx->n)
        return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_window_means: window range outside the series");
    if (count == 0) return KVM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->b_scratch, (size_t)count * 8);
    if (rc) return rc;
    if (ctx->opt_timing) HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(ctx, kvm_launch_window_means(ctx->d_series, ctx->d_prefix, w, (int64_t)first - 1, count,
                                        (double *)ctx->b_scratch.p, ctx->stream));
    if (ctx->opt_timing) HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->b_scratch.p, (size_t)count * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->opt_timing) {  // kernel time alone (the D2H of `count` doubles dominates the call): kvm_last_kernel_ns
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        ctx->last_kernel_ns = (int64_t)(ms * 1e6);
    }
    return KVM_OK;
    });
}

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
static void idx_block_drop() {
    void *p = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_idx_mu);
        p = g_idx_block;
        g_idx_block = nullptr;
        g_idx_cap = 0;
    }
    if (p) hipHostFree(p);
}

static int index_rows_build(kvm_ctx *ctx, int32_t w, bool grouped, kvm_index_rows **out) {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!out) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "null result pointer");
    *out = nullptr;
    if (!ctx->d_series) return fail(ctx, KVM_ERR_NO_SERIES, "no series loaded");
    if (w < 1 || w > 100000 || (int64_t)w > ctx->n)
        return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_index_rows_build: window length outside [1, min(n, EPOCH)]");
    const int64_t count = ctx->n - w + 1;
    if (count > 0x7fffffffLL) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_index_rows_build: series too long");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    kvm_index_dev dev;
    if (ctx->opt_timing) hipEventRecord(ctx->ev0, ctx->stream);
    int rc = kvm_index_intervals_device(ctx, w, count, &dev, ctx->stream);
    if (rc) return fail(ctx, rc, "kvm_index_rows_build: device pass failed (out of device memory?)");
    if (grouped && (rc = kvm_index_group_device(&dev, ctx->stream))) {
        kvm_index_dev_free(&dev);
        return fail(ctx, rc, "kvm_index_rows_build_grouped: grouping pass failed (out of device memory?)");
    }
    if (ctx->opt_timing) hipEventRecord(ctx->ev1, ctx->stream);
    kvm_index_rows *r = new kvm_index_rows();
    const size_t total = (size_t)dev.total, keys_bytes = (total * 8 + 63) & ~(size_t)63, fl_bytes = (total * 4 + 63) & ~(size_t)63;
    const size_t ng = (size_t)dev.n_groups, gk_bytes = (ng * 8 + 63) & ~(size_t)63, go_bytes = ((ng + 1) * 4 + 63) & ~(size_t)63;
    r->block = idx_block_take(keys_bytes + 2 * fl_bytes + gk_bytes + go_bytes + 64, &r->cap);
    if (!r->block) {
        kvm_index_dev_free(&dev);
        delete r;
        return fail(ctx, KVM_ERR_DEVICE, "kvm_index_rows_build: out of pinned host memory");
    }
    char *base = (char *)r->block;
    char *gbase = base + keys_bytes + 2 * fl_bytes;
    const bool ok = hipMemcpyAsync(base, dev.keys, total * 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                    hipMemcpyAsync(base + keys_bytes, dev.first, total * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                    hipMemcpyAsync(base + keys_bytes + fl_bytes, dev.last, total * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                    (ng == 0 || (hipMemcpyAsync(gbase, dev.group_keys, ng * 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                                 hipMemcpyAsync(gbase + gk_bytes, dev.group_off, (ng + 1) * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess)) &&
                    hipStreamSynchronize(ctx->stream) == hipSuccess;
    kvm_index_dev_free(&dev);
    if (!ok) {
        idx_block_give(r->block, r->cap);
        delete r;
        return fail(ctx, KVM_ERR_DEVICE, "kvm_index_rows_build: readback failed");
    }
    if (ctx->opt_timing) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_kernel_ns = (int64_t)(ms * 1e6);
    }
    r->count = (int64_t)total;
    r->keys = (const double *)base;
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

// The reference's phase-2 time model T_2 = A * #intervals + B * (#offsets / 1e5) * m (K/QueryEngine.java:55-57,316-327) with
// the terms this path adds: C, the launch chain of a call, and D, the one unit of work that holds an answer (a query drawn
// from the series always has its self match) and walks the whole window -- m points for the ED engines, one banded DP of
// m macro-steps x O(m) cells (rho = 5 percent of m) for the DTW engines -- and E, the part of a candidate's cost that does not
// grow with m.  The constants are generated: profiles/fit_cost_model.py writes kvm_cost_model.inc next to
// profiles/r03_phase2_cost_model.json (non-negative least squares on relative residuals, 27-48 (intervals, offsets, m) points
// per variant on n = 1e8, early abandon on, selective queries, one MI355X): {C [ms], A [ms per interval], B [ms per 1e5
// offset-points], D, E [ms per 1e5 offsets]}; tests/test_full_size_gpu.py holds the estimate to measured calls.
extern "C" double kvm_phase2_time_estimate_ms(int32_t variant, int64_t disjoint_windows, int64_t candidate_offsets, int32_t length) {
    static const double K[4][5] = {
#include "kvm_cost_model.inc"
    };
    if (variant < 0 || variant > 3) return -1.0;
    const double m = (double)length;
    const double floor_units = variant >= 2 ? m * m / 1e3 : m;
    return K[variant][0] + K[variant][1] * (double)disjoint_windows +
           (K[variant][2] * m + K[variant][4]) * ((double)candidate_offsets / 1e5) + K[variant][3] * floor_units;
}

// The same estimate when the number of candidates that reach the banded DP is known (the library knows it 0.1 ms into a DTW
// call, behind the screens; KVM_STAT_PASS_LB_KEOGH2 of a finished call): the D term above stands for ONE DP -- the lone wave of
// a selective query -- and the DPs of a call run in rounds of the resident waves (three per SIMD: 12 per CU), a last round of
// <= 4 / <= 8 per CU at 0.46 / 0.65 of a full one (kvm_dtw_dp.h).  `dp_count` < 0, or an ED variant: the a-priori form.  With
// the reference's early abandoning the DPs of a long list end early, so for such lists this is an upper bound.
extern "C" double kvm_phase2_time_estimate_dp_ms(int32_t variant, int64_t disjoint_windows, int64_t candidate_offsets, int32_t length,
                                                 int64_t dp_count, int32_t num_cu) {
    const double base = kvm_phase2_time_estimate_ms(variant, disjoint_windows, candidate_offsets, length);
    if (base < 0 || variant < 2 || dp_count < 0) return base;
    static const double K[4][5] = {
#include "kvm_cost_model.inc"
    };
    const double m = (double)length, one_dp = K[variant][3] * m * m / 1e3;
    const int64_t round = 12LL * (num_cu > 0 ? num_cu : 256), tail = dp_count % round;
    double units = (double)(dp_count / round) / 0.46;   // full rounds, in units of a lone DP
    if (tail) units += 3 * tail <= round ? 1.0 : (3 * tail <= 2 * round ? 0.65 / 0.46 : 1.0 / 0.46);
    return base - one_dp + one_dp * units;   // (no DP at all: the call is its chain and its screens)
}

// The phase-1 early-termination test of the four engines (K/QueryEngine.java:316-327: `i >= 5 && totalTimeUsageEstimated >
// lastTotalTimeUsageEstimated`), deterministic: the caller supplies its phase-1 cost so far from a model of its own (index rows
// scanned x a constant, say) instead of reading the wall clock, the phase-2 term is the estimate above.
extern "C" int32_t kvm_phase1_should_stop(int32_t variant, int32_t segment_index, double phase1_ms_so_far, int64_t disjoint_windows,
                                          int64_t candidate_offsets, int32_t length, double *last_total_ms) {
    if (!last_total_ms) return -1;
    const double est = kvm_phase2_time_estimate_ms(variant, disjoint_windows, candidate_offsets, length);
    if (est < 0) return -1;
    const double total = phase1_ms_so_far + est;
    if (segment_index >= 5 && total > *last_total_ms) return 1;   // (`break`: lastTotalTimeUsageEstimated is not updated, :321-324)
    *last_total_ms = total;                                        // :325
    return 0;
}

extern "C" int32_t kvm_resolve_rho(double r, int32_t m) {
    if (r <= 1) return (int32_t)std::floor(r * m);  // K/QueryEngineDtw.java:135-140
    return (int32_t)std::floor(r);
}

extern "C" int kvm_cnsm_dtw_order(const double *query, int32_t m, int32_t *order) {
    return kvm_guard<int>(nullptr, [&]() -> int {
    if (!query || m < 1) return fail(nullptr, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_cnsm_dtw_order: bad arguments");
    // K/NormQueryEngineDtw.java:205-211 (statistics) and :461-464 (z-normalisation)
    double ex = 0, ex2 = 0;
    for (int i = 0; i < m; i++) {
        ex += query[i];
        ex2 += query[i] * query[i];
    }
    const double mean = ex / m, sd = std::sqrt(ex2 / m - mean * mean);
    std::vector<double> zq((size_t)m);
    for (int i = 0; i < m; i++) zq[(size_t)i] = (query[i] - mean) / sd;
    std::vector<int32_t> ord((size_t)m);
    const bool ok = kvm_jsort::cnsm_dtw_order(zq.data(), m, ord.data());
    if (order) std::memcpy(order, ord.data(), (size_t)m * sizeof(int32_t));
    return ok ? KVM_OK : fail(nullptr, KVM_ERR_ILLEGAL_ARGUMENT, "Comparison method violates its general contract!");
    });
}

static bool kvm_trace_on() {
    static int v = -1;
    if (v < 0) v = getenv("KVM_TRACE") ? 1 : 0;
    return v == 1;
}
// KVM_DEBUG_SYNC=1 in the environment: wait for the stream after every stage of a DTW call and name it on stderr (finds a
// kernel that does not return)
static bool kvm_debug_sync_on() {
    static int v = -1;
    if (v < 0) v = getenv("KVM_DEBUG_SYNC") ? 1 : 0;
    return v == 1;
}
#define KVM_DBG_SYNC(ctx, stream, label)                                                                      \
    do {                                                                                                      \
        if (kvm_debug_sync_on()) {                                                                            \
            fprintf(stderr, "[kvm-dbg] %s ...", label);                                                       \
            fflush(stderr);                                                                                   \
            hipError_t e_ = hipStreamSynchronize(stream);                                                     \
            fprintf(stderr, " %s\n", hipGetErrorString(e_));                                                  \
        }                                                                                                     \
    } while (0)
#define KVM_TP(label)                                                                                         \
    do {                                                                                                      \
        if (kvm_trace_on()) {                                                                                 \
            auto now_ = std::chrono::steady_clock::now();                                                     \
            fprintf(stderr, "[kvm] %-22s %8.1f us\n", label,                                                  \
                    std::chrono::duration<double, std::micro>(now_ - tp_).count());                           \
            tp_ = now_;                                                                                       \
        }                                                                                                     \
    } while (0)

// ---- plan: intervals -> runs -> tiles -------------------------------------------------------------------

struct plan {
    int32_t *chunk_lo = nullptr, *chunk_hi = nullptr;  // cNSM-DTW: data range of the chunk each interval is fetched in
    int32_t *left = nullptr, *right = nullptr;  // copies of the interval list in the pinned staging buffer (the runs are
    int64_t n_runs = 0;         // built from them on the GPU, kvm_plan_runs_kernel); n_runs = non-empty clipped intervals
    int64_t n_tiles = 0;        // tiles (class A tiles in hybrid mode)
    int64_t n_tiles_b = 0;      // hybrid mode: class B tiles
    int64_t n_tiles_c = 0;      // hybrid mode: class C tiles
    int64_t cnt_candidate = 0;  // sum(right-left+1) before clipping, K/QueryEngine.java:344
    int64_t windows = 0;        // windows actually evaluated
};

// Totals of the plan over the interval list (the per-run prefix sums are formed on the device from the same rule,
// kvm_plan_interval).  Written as a plain reduction so that the compiler vectorises it; the AVX2 clone is picked at
// run time (15 675 intervals: 31 us scalar, a few us vectorised).
struct plan_totals {
    int64_t cnt_candidate = 0, n_tiles = 0, n_tiles_b = 0, n_tiles_c = 0, windows = 0, nr = 0, bad_rsm = 0;
};
#define PLAN_TOTALS_BODY                                                                                        \
    int64_t cnt = 0, ta = 0, tb = 0, tc = 0, w = 0, nr = 0, bad = 0;                                            \
    for (int32_t p = 0; p < k; p++) {                                                                           \
        const int64_t l = left[p], r = right[p];                                                                \
        cnt += r - l + 1;                                                                                       \
        int64_t lo = l - shift, hi = r - shift;                                                                 \
        lo = lo < 1 ? 1 : lo;                                                                                   \
        hi = hi > hi_max ? hi_max : hi;                                                                         \
        const int64_t c64 = hi - lo + 1;                                                                        \
        bad |= (c64 + m - 1 < 1) ? 1 : 0;                                                                       \
        const int64_t c = c64 > 0 ? c64 : 0;                                                                    \
        const int64_t rem = c & 63;                                                                             \
        const int64_t a_h = (c >> 6) + (rem >= (int64_t)KVM_EDA_MIN ? 1 : 0);                                   \
        const int64_t b_h = (rem > (int64_t)KVM_EDC_TILE && rem < (int64_t)KVM_EDA_MIN) ? 1 : 0;                \
        const int64_t c_h = (rem >= 1 && (rem <= (int64_t)KVM_EDC_TILE ||                                       \
                                          (rem > (int64_t)KVM_EDB_TILE && rem < (int64_t)KVM_EDA_MIN))) ? 1 : 0; \
        const int64_t a_u = (c + ((int64_t)1 << tile_log2) - 1) >> tile_log2;                                   \
        ta += hybrid ? a_h : a_u;                                                                               \
        tb += hybrid ? b_h : 0;                                                                                 \
        tc += hybrid ? c_h : 0;                                                                                 \
        w += c;                                                                                                 \
        nr += c > 0 ? 1 : 0;                                                                                    \
    }                                                                                                           \
    out.cnt_candidate = cnt; out.n_tiles = ta; out.n_tiles_b = tb; out.n_tiles_c = tc; out.windows = w; out.nr = nr; out.bad_rsm = bad;
__attribute__((target("avx2"))) static void plan_totals_avx2(const int32_t *left, const int32_t *right, int32_t k, int64_t shift,
                                                             int64_t hi_max, int64_t m, bool hybrid, int tile_log2,
                                                             plan_totals &out) {
    PLAN_TOTALS_BODY
}
static void plan_totals_generic(const int32_t *left, const int32_t *right, int32_t k, int64_t shift, int64_t hi_max, int64_t m,
                                bool hybrid, int tile_log2, plan_totals &out) {
    PLAN_TOTALS_BODY
}
#undef PLAN_TOTALS_BODY

// Window start offsets of interval [left,right]: max(1,left-shift) .. min(right-shift, n-m+1)
// (K/QueryEngine.java:346-354: begin/end clipping, loop `i + length - 1 < data.size()`).
// `rho_strict` >= 0 (DTW engines with KVM_OPT_STRICT_BOUNDS): a fetched range of <= rho_strict points ends the reference in
// ArrayIndexOutOfBoundsException (lowerUpperLemire over the fetched data, K/QueryEngineDtw.java:399,
// K/NormQueryEngineDtw.java:524 -> K/utils/DtwUtils.java:81-90) -- right after the readTimeSeries that could have thrown
// IllegalArgumentException for the same range, interval by interval (chunk by chunk) in list order.
static int build_plan(kvm_ctx *ctx, bool cnsm, int32_t m, const int32_t *left, const int32_t *right, int32_t k,
                      int32_t shift, int32_t tile_max, bool hybrid, plan &pl, int32_t rho_strict = -1) {
    const int64_t n = ctx->n;
    // The engines hand phase 2 the list sortAndMergeIntervals leaves (K/QueryEngine.java:664-693): sorted by `left`.
    // Lists with overlapping, adjacent, nested, repeated or empty intervals go through like in the Java loops (every
    // interval is scanned on its own: a candidate's records carry the INDEX of its run, two runs may well start at the
    // same offset).  A list that is NOT sorted by `left`: the RSM engines treat intervals independently, so it is sorted
    // here (stable, by `left`) -- the same (offset, distance) pairs come out, in the tie order of the sorted list.  The cNSM
    // engines merge consecutive intervals into one fetched chunk and index it with `left - begin`: out of order that is an
    // IndexOutOfBoundsException or the range check of readTimeSeries in Java, depending on the list; refused here.
    std::vector<int32_t> sorted_l, sorted_r, clipped_r;
    const int32_t *const left_given = left, *const right_given = right;  // (which interval throws first: the list as given)
    {
        bool ascending = true;
        for (int32_t p = 1; p < k; p++) ascending &= left[p] >= left[p - 1];
        if (!ascending) {
            if (cnsm)
                return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "interval list not sorted by left (sortAndMergeIntervals, K/QueryEngine.java:664-693)");
            std::vector<int32_t> idx((size_t)k);
            for (int32_t p = 0; p < k; p++) idx[(size_t)p] = p;
            std::stable_sort(idx.begin(), idx.end(), [left](int32_t a, int32_t b) { return left[a] < left[b]; });
            sorted_l.resize((size_t)k);
            sorted_r.resize((size_t)k);
            for (int32_t p = 0; p < k; p++) {
                sorted_l[(size_t)p] = left[idx[(size_t)p]];
                sorted_r[(size_t)p] = right[idx[(size_t)p]];
            }
            left = sorted_l.data();
            right = sorted_r.data();
        }
    }
    // RSM engines fetch per interval: readTimeSeries(begin, end-begin+1) throws when the clipped range is empty
    // (K/QueryEngine.java:346-352; K/operator/file/TimeSeriesFileOperator.java:55-57).  end - begin + 1 == (clipped window
    // count) + m - 1, so the totals loop below notices an offender (and then words the message from the first one).
    if (cnsm) {
        // cNSM engines fetch greedy-merged chunks of <= MAX_SCAN_DATA_LENGTH points
        // (K/NormQueryEngine.java:454-482); only the chunk read can throw.
        int32_t idx = 0;
        while (idx < k) {
            int32_t beginIdx = idx, endIdx = idx;
            int64_t begin = (int64_t)left[idx] - shift, end = (int64_t)right[idx] - shift + m - 1;
            if (begin < 1) begin = 1;
            int64_t length = end - begin + 1;
            idx++;
            while (idx < k) {
                begin = (int64_t)left[idx] - shift;
                int64_t newLength = length + begin - end - 1;
                end = (int64_t)right[idx] - shift + m - 1;
                if (end > n) end = n;
                newLength += end - begin + 1;
                if (newLength > KVM_MAX_SCAN_DATA_LENGTH) break;
                endIdx = idx;
                length = newLength;
                idx++;
            }
            begin = (int64_t)left[beginIdx] - shift;
            end = (int64_t)right[endIdx] - shift + m - 1;
            if (begin < 1) begin = 1;
            if (end > n) end = n;
            if (end - begin + 1 < 1 || begin > n) {
                char msg[96];
                snprintf(msg, sizeof msg, "left: %lld, length: %lld", (long long)begin, (long long)(end - begin + 1));
                return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, msg);
            }
            if (rho_strict >= 0 && end - begin + 1 <= rho_strict) {
                char msg[32];
                snprintf(msg, sizeof msg, "%lld", (long long)(end - begin + 1 - rho_strict - 1));  // (JDK 8 words it as the index)
                return fail(ctx, KVM_ERR_INDEX_OUT_OF_BOUNDS, msg);
            }
            if (pl.chunk_lo)  // the data envelope of the DTW cascade is built per fetched chunk (K/NormQueryEngineDtw.java:522-524)
                for (int32_t p = beginIdx; p <= endIdx; p++) {
                    pl.chunk_lo[p] = (int32_t)begin;
                    pl.chunk_hi[p] = (int32_t)end;
                }
            // An interval is scanned inside its chunk only: `if (end1 > data.size() - 1) end1 = data.size() - 1`
            // (K/NormQueryEngine.java:494, K/NormQueryEngineDtw.java:536).  The chunk ends where its LAST interval ends, so
            // in a list with overlapping or nested intervals an earlier, longer interval loses the windows that reach past
            // it (found by profiles/fuzz_lists.py); with disjoint sorted intervals the clip never bites.
            for (int32_t p = beginIdx; p <= endIdx; p++) {
                const int64_t last_start = end - m + 1 + shift;  // in the coordinates of `right`
                if ((int64_t)right[p] > last_start) {
                    if (clipped_r.empty()) clipped_r.assign(right, right + k);
                    clipped_r[(size_t)p] = (int32_t)std::max<int64_t>(last_start, (int64_t)left[p] - 1);
                }
            }
        }
    }
    // (cntCandidate is counted from the list as given, K/NormQueryEngine.java:485; the plan from the clipped one)
    int64_t cnt_given = 0;
    const int32_t *right_plan = right;
    if (!clipped_r.empty()) {
        for (int32_t p = 0; p < k; p++) cnt_given += (int64_t)right[p] - left[p] + 1;
        right_plan = clipped_r.data();
    }
    // totals only (launch geometry, buffer sizes): the per-run prefix sums are formed on the device from the same rule
    const int64_t hi_max = n - m + 1;
    int tile_log2 = 0;
    while ((1 << tile_log2) < tile_max) tile_log2++;
    plan_totals tot;
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2) plan_totals_avx2(left, right_plan, k, shift, hi_max, m, hybrid, tile_log2, tot);
    else plan_totals_generic(left, right_plan, k, shift, hi_max, m, hybrid, tile_log2, tot);
    const int64_t cnt_candidate = clipped_r.empty() ? tot.cnt_candidate : cnt_given, n_tiles = tot.n_tiles, n_tiles_b = tot.n_tiles_b, n_tiles_c = tot.n_tiles_c, windows = tot.windows,
                  nr = tot.nr;
    const bool bad_rsm = tot.bad_rsm != 0;
    // (a stub of <= rho < m points holds no window: windows < list length is the cheap sign that one may exist)
    if ((bad_rsm || (rho_strict >= 0 && nr < k)) && !cnsm) {
        for (int32_t p = 0; p < k; p++) {
            int64_t begin = (int64_t)left_given[p] - shift, end = (int64_t)right_given[p] - shift + m - 1;
            if (begin < 1) begin = 1;
            if (end > n) end = n;
            if (end - begin + 1 < 1) {
                char msg[96];
                snprintf(msg, sizeof msg, "left: %lld, length: %lld", (long long)begin, (long long)(end - begin + 1));
                return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, msg);
            }
            if (rho_strict >= 0 && end - begin + 1 <= rho_strict) {
                char msg[32];
                snprintf(msg, sizeof msg, "%lld", (long long)(end - begin + 1 - rho_strict - 1));
                return fail(ctx, KVM_ERR_INDEX_OUT_OF_BOUNDS, msg);
            }
        }
    }
    if (k > 0) {
        std::memcpy(pl.left, left, (size_t)k * 4);
        std::memcpy(pl.right, right_plan, (size_t)k * 4);
    }
    pl.cnt_candidate += cnt_candidate;
    pl.n_tiles += n_tiles;
    pl.n_tiles_b += n_tiles_b;
    pl.n_tiles_c += n_tiles_c;
    pl.windows += windows;
    pl.n_runs = nr;
    return KVM_OK;
}

// Upper/lower envelope of the query: u[i] = max q[max(0,i-r)..min(m-1,i+r)], l[i] = min (same values as
// DtwUtils.lowerUpperLemire, K/utils/DtwUtils.java:50-91, which is a pure selection).  Monotonic deques.
static void query_envelope(const std::vector<double> &q, int r, std::vector<double> &u, std::vector<double> &l) {
    const int m = (int)q.size();
    u.assign(m, 0.0);
    l.assign(m, 0.0);
    std::deque<int> mx, mn;
    int next = 0;
    for (int i = 0; i < m; i++) {
        int hi = std::min(m - 1, i + r), lo = std::max(0, i - r);
        for (; next <= hi; next++) {
            while (!mx.empty() && q[mx.back()] <= q[next]) mx.pop_back();
            mx.push_back(next);
            while (!mn.empty() && q[mn.back()] >= q[next]) mn.pop_back();
            mn.push_back(next);
        }
        while (mx.front() < lo) mx.pop_front();
        while (mn.front() < lo) mn.pop_front();
        u[i] = q[mx.front()];
        l[i] = q[mn.front()];
    }
}

// ---- the shared verification driver ---------------------------------------------------------------------

struct verify_out {
    int32_t *d_off;
    double *d_dist;
    int64_t cap;
    bool own;  // true: answers go to the context's own buffers, sized by run_variant once the plan knows #windows
    // cNSM-DTW with KVM_OPT_UCR_LOOPS: what the sequential tie path needs to reconstruct the executor's never-cleared cb1 / cb2
    kvm_replay_ws rws = {};
    const kvm_tile *tiles = nullptr;
    int64_t n_tiles = 0, windows = 0;
    bool ucr_tables_ready = false;
};

enum { V_RSM_ED = 0, V_CNSM_ED = 1, V_RSM_DTW = 2, V_CNSM_DTW = 3 };

// The envelope is built on the stream of the call that needs it (its consumers run on that stream).  A cached envelope
// may have been built on ANOTHER stream by an earlier kvm_verify_device call that returned without synchronising: the
// event recorded behind the build orders this call's stream after it.
static int ensure_envelope(kvm_ctx *ctx, int32_t rho, hipStream_t stream) {
    if (ctx->env_rho == rho && ctx->d_env_u) {
        if (ctx->ev_env && ctx->env_stream != stream) HIPCHK(ctx, hipStreamWaitEvent(stream, ctx->ev_env, 0));
        return KVM_OK;
    }
    if (!ctx->d_env_u) {
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_env_u, (size_t)ctx->n * 8));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_env_l, (size_t)ctx->n * 8));
    }
    ctx->env_rho = -1;
    // (a rebuild for another rho overwrites arrays the previous call's kernels may still be reading on another stream:
    // run_variant has already waited for that call's ev_staged, which is recorded behind all of them)
    HIPCHK(ctx, kvm_launch_envelope(ctx->d_series, ctx->n, rho, ctx->d_env_u, ctx->d_env_l, stream));
    if (!ctx->ev_env) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_env, hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(ctx->ev_env, stream));
    ctx->env_stream = stream;
    ctx->env_rho = rho;
    return KVM_OK;
}

// Upper / lower envelope of radius rho of the resident series, points first .. first + count - 1 (1-based): what
// DtwUtils.lowerUpperLemire (K/utils/DtwUtils.java:93-134) returns for the whole series fetched as one chunk.  The DTW
// engines use it for every point that is at least rho away from both ends of its fetched chunk.
extern "C" int kvm_data_envelope(kvm_ctx *ctx, int32_t rho, int32_t first, int64_t count, double *upper, double *lower) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!ctx->d_series) return fail(ctx, KVM_ERR_NO_SERIES, "no series loaded");
    if (rho < 0 || first < 1 || count < 0 || (int64_t)first + count - 1 > ctx->n || !upper || !lower)
        return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "bad kvm_data_envelope arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_envelope(ctx, rho, ctx->stream);
    if (rc) return rc;
    if (count > 0) {
        HIPCHK(ctx, hipMemcpyAsync(upper, ctx->d_env_u + (first - 1), (size_t)count * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(lower, ctx->d_env_l + (first - 1), (size_t)count * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return KVM_OK;
    });
}

// Runs the kernels of one variant; survivors land (unsorted) in out.d_off/out.d_dist, counters stay on the
// device in ctx->b_counters.  `stream` is the stream to launch on.
static int run_variant(kvm_ctx *ctx, int variant, const double *query, int32_t m, const int32_t *left,
                       const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho, double alpha,
                       double beta, plan &pl, verify_out &out, hipStream_t stream, int64_t *kernel_ns,
                       kvm_query_dev &dtw_qd, kvm_dtw_aux &dtw_ax) {
    if (!ctx->d_series) return fail(ctx, KVM_ERR_NO_SERIES, "no series loaded");
    if (!query || m < 1 || k < 0 || (k > 0 && (!left || !right)))
        return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "bad query/interval arguments");
    if (m > ctx->n) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "query longer than the series");
    if (!(epsilon >= 0)) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "epsilon must be >= 0");
    const bool cnsm = variant == V_CNSM_ED || variant == V_CNSM_DTW;
    const bool dtw = variant == V_RSM_DTW || variant == V_CNSM_DTW;
    // (the REPLs refuse `alpha <= 0.0` (K/NormQueryEngine.java:139-142); a NaN passes that test, and in the engine neither
    //  `std / stdQ <= alpha` nor `>= 1 / alpha` holds for it (:511): no answers -- the guarded filters keep every candidate
    //  (a NaN compares false) and the exact (alpha, beta) test, which is the reference's own, drops them all)
    if (cnsm && alpha <= 0) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "alpha must be > 0");
    if (dtw && rho < 0) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "rho must be >= 0");
    // ---- query-side constants (host, sequential, exactly the reference's arithmetic) ----
    std::vector<double> q(query, query + m), qu, ql;
    kvm_query_dev qd;
    std::memset(&qd, 0, sizeof qd);
    qd.m = m;
    qd.rho = rho;
    qd.eps2 = epsilon * epsilon;
    qd.normalised = cnsm ? 1 : 0;
    qd.early_abandon = (int32_t)ctx->opt_early_abandon;
    qd.alpha = alpha;
    qd.beta = beta;
    if (cnsm) {
        // K/NormQueryEngine.java:192-198 (phase 0) and :438-441 (z-normalisation)
        double ex = 0, ex2 = 0;
        for (int i = 0; i < m; i++) {
            ex += query[i];
            ex2 += query[i] * query[i];
        }
        qd.meanQ = ex / m;
        qd.stdQ = std::sqrt(ex2 / m - qd.meanQ * qd.meanQ);
        for (int i = 0; i < m; i++) q[i] = (query[i] - qd.meanQ) / qd.stdQ;
    }
    // DtwUtils.lowerUpperLemire(Q, length, rho, l, u) (K/QueryEngineDtw.java:362, K/NormQueryEngineDtw.java:469) stores
    // `u[i - r - 1]` from i = len on (K/utils/DtwUtils.java:81-90): with rho >= m that is ArrayIndexOutOfBoundsException
    // before anything has been read -- and before the sort below.  KVM_OPT_STRICT_BOUNDS = 0 computes the natural extension.
    if (dtw && ctx->opt_strict_bounds && m <= rho) {
        char msg[32];
        snprintf(msg, sizeof msg, "%d", (int)(m - rho - 1));
        return fail(ctx, KVM_ERR_INDEX_OUT_OF_BOUNDS, msg);
    }
    if (dtw) query_envelope(q, rho, qu, ql);
    // The reference's summation order of the LB_Keogh loops.  RSM-DTW: the reorder is commented out, order = identity
    // (K/QueryEngineDtw.java:364-383).  cNSM-DTW: Arrays.sort with the comparator (int)(|o2| - |o1|)
    // (K/NormQueryEngineDtw.java:475-478) -- inconsistent, so the permutation is the one JDK 8's TimSort leaves
    // (kvm_jsort.h), and the sort may fail the way Java throws: query() then ends in that IllegalArgumentException before
    // it has read anything, so this comes first here too.  KVM_OPT_SORT_VIOLATION = 1 answers such a query anyway,
    // summing in plain index order (the answers differ from a successful sort's only at exact ties with eps^2).
    std::vector<int32_t> ord;
    if (dtw) {
        ord.resize((size_t)m);
        for (int i = 0; i < m; i++) ord[(size_t)i] = i;
        if (cnsm && !kvm_jsort::cnsm_dtw_order(q.data(), m, ord.data())) {
            if (!ctx->opt_sort_violation)
                return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "Comparison method violates its general contract!");
            for (int i = 0; i < m; i++) ord[(size_t)i] = i;
        }
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // Scratch buffers and the pinned staging area are reused by every call: wait until the previous call's device
    // work has drained (kvm_verify_device returns without synchronising the caller's stream).
    if (ctx->done_pending) {
        // the previous kvm_verify_device call may still be reading the device scratch on its stream: this call's stream waits
        // for it on the DEVICE (a call on the same stream is ordered behind it anyway)
        if (ctx->done_stream != stream) HIPCHK(ctx, hipStreamWaitEvent(stream, ctx->ev_done, 0));
        ctx->done_pending = false;
    }
    if (ctx->staged_pending) {
        HIPCHK(ctx, hipEventSynchronize(ctx->ev_staged));
        ctx->staged_pending = false;
    }
    // ED pruning kernel: 1 = wave-per-candidate (the cross-check), 3 = LDS-tiled correlation form; 0 = auto (KVM_ED_KERNEL in
    // the environment overrides auto).  (2, the direct LDS-tiled form, and 4, lane-granular tiles staged by LDS-DMA, were
    // removed in rounds 3 and 4: the latter stayed 2 - 3 % behind the default -- HISTORY.md.)
    static const int env_ed_kernel = [] {
        const char *e = getenv("KVM_ED_KERNEL");
        const int v = e ? atoi(e) : 0;
        return (v == 1 || v == 3) ? v : 0;
    }();
    const int ed_kernel = ctx->opt_ed_kernel != 0 ? (int)ctx->opt_ed_kernel : (env_ed_kernel ? env_ed_kernel : 3);
    auto tp_ = std::chrono::steady_clock::now();
    // 32-candidate tiles for the LDS-tiled kernels (ED pruners, fused DTW lower-bound stage), 64 otherwise
    // (DTW cascade: 64-candidate tiles first, trimmed and cut to <= 32 on the device by kvm_retile_dtw_kernel)
    // cNSM-ED with the dot pruner: uniform 64-candidate tiles first, trimmed and classified on the device by the filter;
    // RSM-ED with early abandoning: the same, trimmed by the head of the window (kvm_retile_rsm_kernel)
    const bool retile_rsm = !cnsm && !dtw && ed_kernel == 3 && ctx->opt_early_abandon;
    const bool retile = (cnsm && !dtw && ed_kernel == 3) || retile_rsm;
    // host-planned list for the correlation-form pruner (RSM-ED, every point evaluated): the class A / B / C tiling
    // (kvm_ed_dot.hip)
    const int32_t tile_max = retile ? KVM_TILE_MAX : (!dtw && ed_kernel == 3) ? KVM_TILE_TILED : KVM_TILE_MAX;
    const bool hybrid = !dtw && ed_kernel == 3 && !retile;
    // query block: q | (qu | ql for DTW) | pad || (qc | pad | q_pref | q_pref8 | qT for ED) || (zQ sorted | order: cNSM-ED
    // exact pass); the runs follow it in the pinned staging buffer, where build_plan writes them directly
    const int nchunks4 = (m + ED4_JC - 1) / ED4_JC, nchunks8 = (m + ED8_JC - 1) / ED8_JC;
    const size_t qmain = (size_t)m * 8 * (dtw ? 3 : 1) + KVM_SERIES_PAD * sizeof(double);
    const size_t qdot = dtw ? 0 : (size_t)m * 8 + KVM_SERIES_PAD * sizeof(double) + ((size_t)2 * nchunks4 + 1 + nchunks8 + 1) * 16;
    const size_t qexact_off = qmain + qdot;
    // DTW: the reference's summation order of the LB_Keogh loops and the query / its envelope in that order
    // (uo | lo | qo | order) behind the main block
    const size_t qbytes = (qexact_off + ((cnsm && !dtw) ? (size_t)m * 12 : 0) + (dtw ? (size_t)m * 28 : 0) + 15) / 16 * 16;
    // the interval list follows the query block (left | right, each padded to 16 B; cNSM-DTW: + the chunk bounds per
    // interval); the runs built from it on the device follow that
    const bool chunks = cnsm && dtw;
    const size_t lr_half = ((size_t)(k > 0 ? k : 0) * 4 + 15) / 16 * 16, lr_bytes = (chunks ? 4 : 2) * lr_half;
    int rc = ensure_pinned(ctx, qbytes + lr_bytes + 64);
    if (rc) return rc;
    char *hp = (char *)ctx->h_pinned;
    pl.left = (int32_t *)(hp + qbytes);
    pl.right = (int32_t *)(hp + qbytes + lr_half);
    if (chunks) {
        pl.chunk_lo = (int32_t *)(hp + qbytes + 2 * lr_half);
        pl.chunk_hi = (int32_t *)(hp + qbytes + 3 * lr_half);
    }
    rc = build_plan(ctx, cnsm, m, left, right, k, shift, tile_max, hybrid, pl, (dtw && ctx->opt_strict_bounds) ? rho : -1);
    if (rc) return rc;
    // Class-C tiles (two <= 16-candidate remainders per row, kvm_ed_dot.hip) fill the lanes better but a class-C unit
    // lasts 1.3 x as long as a class-B unit: always worth it when the list needs more than one round of the chip's wave
    // slots (throughput); below that the launcher's model of the launch decides (kvm_ed_dot_choose_classes).  Decided
    // here, from the totals; the device-side plan follows the same rule (kvm_plan_tiles: hybrid = 2 with class C, 1 without).
    int hybrid_mode = 0;
    if (hybrid) {
        hybrid_mode = kvm_ed_dot_choose_classes(m, pl.n_tiles, pl.n_tiles_b, pl.n_tiles_c, ctx->num_cu);
        if (const char *e = getenv("KVM_ED_HYBRID"))  // experiments: force the tiling (1 = classes A / B, 2 = A / B / C)
            if (e[0] == '1' || e[0] == '2') hybrid_mode = e[0] - '0';
        if (hybrid_mode == 1) {
            pl.n_tiles_b += pl.n_tiles_c;
            pl.n_tiles_c = 0;
        }
    }
    if (out.own) {  // capacity: every evaluated window may be an answer
        if ((rc = ensure(ctx, ctx->b_out_off, (size_t)pl.windows * 4 + 64))) return rc;
        if ((rc = ensure(ctx, ctx->b_out_dist, (size_t)pl.windows * 8 + 64))) return rc;
        out.d_off = (int32_t *)ctx->b_out_off.p;
        out.d_dist = (double *)ctx->b_out_dist.p;
        out.cap = pl.windows;
    }
    KVM_TP("  build_plan");

    // The interval list goes to the device first and the runs are built from it there (kvm_plan_runs_kernel, ~12 us)
    // while the host prepares the query block below.  Device layout: query block | left | right | runs | plan scratch.
    const size_t rbytes = (size_t)pl.n_runs * sizeof(kvm_run);
    const size_t plan_scratch = ((size_t)(k > 0 ? k : 0) / 256 + 1) * KVM_PLAN_SUMS_BYTES;  // block totals of kvm_launch_plan_runs
    const size_t rc_off = qbytes + lr_bytes + ((rbytes + 31) & ~(size_t)31) + ((plan_scratch + 31) & ~(size_t)31);
    const size_t hist_off = (rc_off + (chunks ? (size_t)pl.n_runs * sizeof(int2) : 0) + 15) & ~(size_t)15;  // cNSM: run_hist
    if ((rc = ensure(ctx, ctx->b_query, hist_off + (cnsm ? (size_t)pl.n_runs * sizeof(double) : 0) + 64))) return rc;
    kvm_run *d_runs_w = (kvm_run *)((char *)ctx->b_query.p + qbytes + lr_bytes);
    int2 *d_run_chunk = chunks ? (int2 *)((char *)ctx->b_query.p + rc_off) : nullptr;
    double *d_run_hist = cnsm ? (double *)((char *)ctx->b_query.p + hist_off) : nullptr;
    qd.run_hist = d_run_hist;
    if (rbytes) {
        int tile_log2 = 0;
        while ((1 << tile_log2) < tile_max) tile_log2++;
        // (the first plan kernel reads the interval list straight from the pinned staging buffer and leaves the device copy)
        HIPCHK(ctx, kvm_launch_plan_runs(pl.left, pl.right, (int32_t *)((char *)ctx->b_query.p + qbytes),
                                         (int32_t *)((char *)ctx->b_query.p + qbytes + lr_half), k, shift,
                                         ctx->n - m + 1, hybrid_mode, tile_log2, d_runs_w,
                                         (char *)ctx->b_query.p + qbytes + lr_bytes + ((rbytes + 31) & ~(size_t)31), stream,
                                         pl.chunk_lo, pl.chunk_hi, d_run_chunk));
    }

    KVM_TP("  plan launches");
    // correlation-form ED pruner: centred query and its chunk-granular prefix sums
    std::vector<double> qcen;
    std::vector<double2> qpref, qpref8;
    if (!dtw) {
        // (pruner-side quantities, guard-banded: the summation order is free, so four interleaved partial sums keep the
        // adders busy -- the plain sequential loops cost ~20 us of the host's critical path at m = 8192)
        double s4[4] = {0, 0, 0, 0};
        int i4 = 0;
        for (; i4 + 4 <= m; i4 += 4)
            for (int u = 0; u < 4; u++) s4[u] += q[i4 + u];
        for (; i4 < m; i4++) s4[0] += q[i4];
        qd.q_mean = ((s4[0] + s4[1]) + (s4[2] + s4[3])) / m;
        qcen.resize(m);
        qpref.resize((size_t)2 * nchunks4 + 1);  // ED16_JC-granular: class C uses every entry, class B every second one
        qpref8.resize((size_t)nchunks8 + 1);
        static_assert(ED4_JC % 4 == 0 && ED8_JC % 4 == 0 && ED16_JC % 4 == 0 && ED4_JC == 2 * ED16_JC,
                      "chunk boundaries fall between groups of four; class B chunks are two class C chunks");
        double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
        const double qm = qd.q_mean;
        int i = 0;
        for (; i < m; i += 4) {
            if (i % ED16_JC == 0) qpref[(size_t)(i / ED16_JC)] = make_double2((a1[0] + a1[1]) + (a1[2] + a1[3]), (a2[0] + a2[1]) + (a2[2] + a2[3]));
            if (i % ED8_JC == 0) qpref8[(size_t)(i / ED8_JC)] = make_double2((a1[0] + a1[1]) + (a1[2] + a1[3]), (a2[0] + a2[1]) + (a2[2] + a2[3]));
            const int lim = std::min(4, m - i);
            for (int u = 0; u < lim; u++) {
                const double c = q[i + u] - qm;
                qcen[i + u] = c;
                a1[u] += c;
                a2[u] += c * c;
            }
        }
        const double p1 = (a1[0] + a1[1]) + (a1[2] + a1[3]), p2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
        for (size_t t = (size_t)((m + ED16_JC - 1) / ED16_JC); t <= (size_t)2 * nchunks4; t++) qpref[t] = make_double2(p1, p2);
        qpref8[(size_t)nchunks8] = make_double2(p1, p2);
    }
    const size_t tbytes = (size_t)((retile || dtw) ? 3 * pl.n_tiles : pl.n_tiles + pl.n_tiles_b + pl.n_tiles_c) * sizeof(kvm_tile);
    // (KVM_SERIES_PAD spare doubles after q: the tiled ED kernel stages whole query chunks without bounds checks)
    if ((rc = ensure(ctx, ctx->b_tiles, tbytes + 64))) return rc;
    if ((rc = ensure(ctx, ctx->b_counters, KVM_CNT_TOTAL * sizeof(unsigned long long)))) return rc;
    // zero pads: after q (| qu | ql), after qc, and the tail of qT
    std::memset(hp + (size_t)m * 8 * (dtw ? 3 : 1), 0, KVM_SERIES_PAD * sizeof(double));
    if (!dtw) std::memset(hp + qmain + (size_t)m * 8, 0, KVM_SERIES_PAD * sizeof(double));
    std::memcpy(hp, q.data(), (size_t)m * 8);
    if (!dtw) {
        std::memcpy(hp + qmain, qcen.data(), (size_t)m * 8);
        std::memcpy(hp + qmain + (size_t)m * 8 + KVM_SERIES_PAD * sizeof(double), qpref.data(), qpref.size() * 16);
        std::memcpy(hp + qmain + (size_t)m * 8 + KVM_SERIES_PAD * sizeof(double) + qpref.size() * 16, qpref8.data(),
                    qpref8.size() * 16);
    }
    const size_t qfirst = (cnsm && !dtw) ? qexact_off : qbytes;  // the exact-pass block is staged later (see below)
    if (dtw) {
        std::memcpy(hp + (size_t)m * 8, qu.data(), (size_t)m * 8);
        std::memcpy(hp + (size_t)m * 16, ql.data(), (size_t)m * 8);
        double *h_uo = (double *)(hp + qexact_off), *h_lo = h_uo + m, *h_qo = h_lo + m;
        int32_t *h_order = (int32_t *)(h_qo + m);
        for (int i = 0; i < m; i++) {
            const int o = ord[(size_t)i];
            h_order[i] = o;
            h_uo[i] = qu[(size_t)o];
            h_lo[i] = ql[(size_t)o];
            h_qo[i] = q[(size_t)o];
        }
    }
    KVM_TP("  query prep + staging");
    // scratch of the running-sum replay of the exact pass (cNSM): flags + {mean,std} per window, counts per run
    kvm_replay_ws rws;
    std::memset(&rws, 0, sizeof rws);
    if (cnsm && pl.windows > 0) {
        const size_t sbytes = (size_t)pl.windows * sizeof(double2);
        const size_t mbytes = (size_t)pl.n_runs * sizeof(int32_t);
        if ((rc = ensure(ctx, ctx->b_scratch, sbytes + 2 * mbytes + 64))) return rc;
        rws.runs = (const kvm_run *)((const char *)ctx->b_query.p + qbytes + lr_bytes);
        rws.n_runs = pl.n_runs;
        rws.stats = (double *)ctx->b_scratch.p;
        rws.stats_plane = pl.windows;
        rws.run_maxk = (int32_t *)((char *)ctx->b_scratch.p + sbytes);
        rws.run_list = rws.run_maxk + pl.n_runs;
        rws.run_list_n = (unsigned long long *)ctx->b_counters.p + KVM_CNT_REPLAY_RUNS;
    }
    // the query block (the tile list -- 16 B per tile -- is expanded from the runs on the GPU, by the kernel that also
    // zeroes the counters; for cNSM-ED the exact-pass block is filled in and copied further down, in stream order, before
    // its only reader -- the exact pass -- is launched)
    // (the upload rides on the expansion launch: one launch less per call -- a latency-bound call is a chain of launches)
    const kvm_run *d_runs = d_runs_w;
    if (rbytes) {
        if (hybrid)
            HIPCHK(ctx, kvm_launch_expand_tiles_hybrid(d_runs, pl.n_runs, hybrid_mode, (kvm_tile *)ctx->b_tiles.p,
                                                       (kvm_tile *)ctx->b_tiles.p + pl.n_tiles,
                                                       (kvm_tile *)ctx->b_tiles.p + pl.n_tiles + pl.n_tiles_b,
                                                       (unsigned long long *)ctx->b_counters.p, rws.run_maxk, stream,
                                                       ctx->b_query.p, hp, qfirst));
        else
            // (cNSM: also what the reference's running sums of each interval pass over -- guard of the pruners, kvm_hist_*)
            HIPCHK(ctx, kvm_launch_expand_tiles(d_runs, pl.n_runs, tile_max, (kvm_tile *)ctx->b_tiles.p,
                                                (unsigned long long *)ctx->b_counters.p, rws.run_maxk, stream,
                                                ctx->d_blockmax, d_run_hist, m, ctx->n, ctx->b_query.p, hp, qfirst));
    } else {
        HIPCHK(ctx, kvm_launch_upload(ctx->b_query.p, hp, qfirst, stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->b_counters.p, 0, KVM_CNT_TOTAL * sizeof(unsigned long long), stream));
    }
    qd.q = (const double *)ctx->b_query.p;
    qd.qc = dtw ? nullptr : (const double *)((const char *)ctx->b_query.p + qmain);
    qd.q_pref = dtw ? nullptr
                    : (const double2 *)((const char *)ctx->b_query.p + qmain + (size_t)m * 8 +
                                        KVM_SERIES_PAD * sizeof(double));
    qd.q_pref8 = dtw ? nullptr : qd.q_pref + (2 * nchunks4 + 1);
    qd.qu = dtw ? qd.q + m : nullptr;
    qd.ql = dtw ? qd.q + 2 * (size_t)m : nullptr;
    unsigned long long *counters = (unsigned long long *)ctx->b_counters.p;
    const kvm_tile *d_tiles = (const kvm_tile *)ctx->b_tiles.p;
    const int64_t n_tiles = pl.n_tiles;

    if (kernel_ns) *kernel_ns = 0;
    ctx->last_kernel_ns = 0;
    ctx->timing_pending = false;
    KVM_TP("  H2D enqueue");
    if (n_tiles + pl.n_tiles_b + pl.n_tiles_c == 0) return KVM_OK;

    // (the default ED pruner on a host-planned list stamps its own begin and end: kvm_launch_ed_dot)
    const bool self_timed = ctx->opt_timing && !dtw && !retile && ed_kernel == 3;
    if (ctx->opt_timing && !self_timed) HIPCHK(ctx, hipEventRecord(ctx->ev0, stream));
    if (!dtw) {
        // pruning pass (fast kernels, guard band) -> survivors -> exact pass in the reference's operation order
        if ((rc = ensure(ctx, ctx->b_cands, (size_t)pl.windows * sizeof(kvm_surv) + 64))) return rc;
        kvm_surv *surv = (kvm_surv *)ctx->b_cands.p;
        if (retile) {
            kvm_tile *tA = (kvm_tile *)ctx->b_tiles.p + pl.n_tiles, *tB = tA + pl.n_tiles;
            if (retile_rsm)
                HIPCHK(ctx, kvm_launch_retile_rsm(ctx->d_series, qd, d_tiles, pl.n_tiles, tA, tB, counters + KVM_CNT_TILES_A,
                                                  stream));
            else
                HIPCHK(ctx, kvm_launch_retile_cnsm(ctx->d_series, ctx->d_prefix, qd, d_tiles, pl.n_tiles, tA, tB,
                                                   counters + KVM_CNT_TILES_A, stream));
            KVM_TP("  retile launch");
            HIPCHK(ctx, kvm_launch_ed_dot(ctx->d_series, ctx->d_prefix, ctx->n, qd, tA, pl.n_tiles, tB, pl.n_tiles, tB, 0, surv,
                                          pl.windows, counters, ctx->num_cu, counters + KVM_CNT_TILES_A, out.d_off, out.d_dist, out.cap,
                                          stream));
            KVM_TP("  ed_dot launch");
        }
        else if (ed_kernel == 3)
            HIPCHK(ctx, kvm_launch_ed_dot(ctx->d_series, ctx->d_prefix, ctx->n, qd, d_tiles, pl.n_tiles, d_tiles + pl.n_tiles,
                                          pl.n_tiles_b, d_tiles + pl.n_tiles + pl.n_tiles_b, pl.n_tiles_c, surv, pl.windows, counters, ctx->num_cu, nullptr, out.d_off, out.d_dist,
                                          out.cap, stream, self_timed ? ctx->ev0 : nullptr, self_timed ? ctx->ev1 : nullptr));
        else
            HIPCHK(ctx, kvm_launch_ed_wave(ctx->d_series, ctx->d_prefix, ctx->n, qd, d_tiles, n_tiles, surv,
                                           pl.windows, counters, stream));
        if (ctx->opt_timing && !self_timed) HIPCHK(ctx, hipEventRecord(ctx->ev1, stream));
        if (cnsm) {
            // cNSM-ED exact pass: the reference sums over the |zQ|-descending permutation (stable sort with
            // Double.compare(|o2|, |o1|), K/NormQueryEngine.java:443-452).  Computed here, while the pruning kernel runs.
            // Stable LSD radix sort on the bit patterns: for |x| the IEEE-754 pattern is monotone, NaN (canonicalised as
            // Double.compare does) sorts above +Inf, and complementing the key turns ascending into descending.
            double *zqs = (double *)(hp + qexact_off);
            int32_t *order = (int32_t *)(hp + qexact_off + (size_t)m * 8);
            {
                std::vector<uint64_t> ka((size_t)m), kb((size_t)m);
                std::vector<int32_t> ia((size_t)m), ib((size_t)m);
                for (int i = 0; i < m; i++) {
                    double a = std::fabs(q[i]);
                    uint64_t u;
                    std::memcpy(&u, &a, 8);
                    if (a != a) u = 0x7ff8000000000000ULL;
                    ka[(size_t)i] = ~u;
                    ia[(size_t)i] = i;
                }
                for (int pass = 0; pass < 8; pass++) {
                    const int sh = pass * 8;
                    size_t hist[257] = {0};
                    for (int i = 0; i < m; i++) hist[((ka[(size_t)i] >> sh) & 255u) + 1]++;
                    bool one_bucket = false;
                    for (int b = 1; b <= 256; b++) one_bucket |= hist[b] == (size_t)m;
                    if (one_bucket) continue;  // every key has the same digit: the pass would be the identity
                    for (int b = 0; b < 256; b++) hist[b + 1] += hist[b];
                    for (int i = 0; i < m; i++) {
                        const size_t d = hist[(ka[(size_t)i] >> sh) & 255u]++;
                        kb[d] = ka[(size_t)i];
                        ib[d] = ia[(size_t)i];
                    }
                    ka.swap(kb);
                    ia.swap(ib);
                }
                for (int i = 0; i < m; i++) {
                    order[i] = ia[(size_t)i];
                    zqs[i] = q[order[i]];
                }
            }
            HIPCHK(ctx, kvm_launch_upload((char *)ctx->b_query.p + qexact_off, hp + qexact_off, (size_t)m * 12, stream));
        }
        KVM_TP("  cnsm order + upload");
        const double *q_exact = cnsm ? (const double *)((const char *)ctx->b_query.p + qexact_off) : qd.q;
        const int32_t *d_order =
            cnsm ? (const int32_t *)((const char *)ctx->b_query.p + qexact_off + (size_t)m * 8) : nullptr;
        // KVM_OPT_UCR_LOOPS (executor mirrors only): UcrEdQueryExecutor's `sum < bsf` loop -- the windows whose ordered partial
        // sums may land exactly on eps^2 go through the literal loop behind the survivors (kvm_ucr.hip)
        kvm_surv *ties = nullptr;
        if (cnsm && ctx->opt_ucr_loops) {
            if ((rc = ensure(ctx, ctx->b_cands2, (size_t)pl.windows * sizeof(kvm_surv) + 64))) return rc;
            ties = (kvm_surv *)ctx->b_cands2.p;
            if (retile) {
                const kvm_tile *tA = (const kvm_tile *)ctx->b_tiles.p + pl.n_tiles, *tB = tA + pl.n_tiles;
                HIPCHK(ctx, kvm_launch_ucr_ed_screen(ctx->d_series, ctx->d_prefix, qd, q_exact, d_order, tA, pl.n_tiles, tB, pl.n_tiles,
                                                     counters + KVM_CNT_TILES_A, ties, pl.windows, counters, ctx->num_cu, stream));
            } else {
                HIPCHK(ctx, kvm_launch_ucr_ed_screen(ctx->d_series, ctx->d_prefix, qd, q_exact, d_order, d_tiles, n_tiles, nullptr, 0,
                                                     nullptr, ties, pl.windows, counters, ctx->num_cu, stream));
            }
        }
        HIPCHK(ctx, kvm_launch_exact_ed(ctx->d_series, qd, q_exact, d_order, surv, counters + KVM_CNT_CANDS1, pl.windows,
                                        rws, out.d_off, out.d_dist, out.cap, counters, ctx->num_cu, stream, ties,
                                        counters + KVM_CNT_UCR_TIES, ties ? pl.windows : 0));
        KVM_TP("  exact launch");
    } else {
        const int use_lb = (int)ctx->opt_lb_cascade;
        if ((rc = ensure(ctx, ctx->b_cands, (size_t)pl.windows * sizeof(kvm_cand) + 64))) return rc;
        if ((rc = ensure(ctx, ctx->b_cands2, (size_t)pl.windows * sizeof(kvm_cand) + 64))) return rc;  // ambiguous candidates
        if ((rc = ensure_envelope(ctx, rho, stream))) return rc;
        if ((rc = ensure(ctx, ctx->b_aux, (size_t)kvm_dtw_dp_max_waves(ctx->num_cu) * 8 * (size_t)rho * sizeof(double) + 64))) return rc;
        kvm_cand *c1 = (kvm_cand *)ctx->b_cands.p;
        kvm_dtw_aux &ax = dtw_ax;
        const double *d_dtw = (const double *)((const char *)ctx->b_query.p + qexact_off);
        ax.uo = d_dtw;
        ax.lo = d_dtw + m;
        ax.qo = d_dtw + 2 * (size_t)m;
        ax.order = (const int32_t *)(d_dtw + 3 * (size_t)m);
        ax.env_u = ctx->d_env_u;
        ax.env_l = ctx->d_env_l;
        ax.runs = d_runs_w;
        ax.n_runs = pl.n_runs;
        ax.run_chunk = d_run_chunk;
        ax.patch = (double *)ctx->b_aux.p;
        ax.ambig = (kvm_cand *)ctx->b_cands2.p;
        ax.cb_exact = nullptr;
        // a wave-parallel sum of m terms against the same terms added one by one (1e-12 + 5e-16 m), plus -- cNSM -- terms
        // formed with a reciprocal instead of a division per term (<= 6u m sqrt(sum), see dp_term_q in kvm_dtw.hip)
        ax.tau = 1e-12 + 5e-16 * (double)m + (cnsm ? 1e-15 * (double)m / std::sqrt(qd.eps2 > 0 ? qd.eps2 : 1.0) : 0.0);
        KVM_DBG_SYNC(ctx, stream, "plan / upload / envelope");
        if (use_lb) {
            // (alpha, beta) filter + LB_Kim for every candidate, survivors re-tiled on the device; then LB_Keogh on the
            // query envelope, in the reference's order, as a screen
            kvm_tile *t2 = (kvm_tile *)ctx->b_tiles.p + n_tiles;
            HIPCHK(ctx, kvm_launch_retile_dtw(ctx->d_series, ctx->d_prefix, qd, d_tiles, n_tiles, t2, counters, stream));
            KVM_DBG_SYNC(ctx, stream, "retile_dtw");
            HIPCHK(ctx, kvm_launch_dtw_screen(ctx->d_series, ctx->d_prefix, qd, ax, t2, n_tiles, counters + KVM_CNT_TILES_A, c1,
                                              counters, stream));
            KVM_DBG_SYNC(ctx, stream, "dtw_screen");
        } else {
            HIPCHK(ctx, kvm_launch_dtw_filter(ctx->d_series, ctx->d_prefix, ctx->n, qd, d_tiles, n_tiles, 0, c1, counters,
                                              stream));
        }
        const unsigned long long *dp_n = counters + KVM_CNT_CANDS1;
        // cNSM: replay the reference's running sums for the candidates that reach the deciding stage (exact mean/std and
        // exact (alpha, beta) decision), so that every value it forms is the reference's own
        if (cnsm)
            HIPCHK(ctx, kvm_launch_exact_stats(ctx->d_series, qd, c1, dp_n, pl.windows, rws, counters, ctx->num_cu, stream));
        KVM_DBG_SYNC(ctx, stream, "exact_stats");
        if ((rc = ensure(ctx, ctx->b_plans, kvm_dtw_plan_bytes(pl.windows) + 64))) return rc;
        HIPCHK(ctx, kvm_launch_dtw_dp(ctx->d_series, ctx->n, qd, ax, c1, dp_n, pl.windows, ctx->b_plans.p, out.d_off, out.d_dist,
                                      out.cap, counters, ctx->num_cu, (int)ctx->opt_dp_kernel, stream,
                                      ctx->opt_timing ? ctx->ev0 : nullptr));  // (timing: the DP kernel alone)
        KVM_DBG_SYNC(ctx, stream, "dtw_dp");
        dtw_qd = qd;
        if (cnsm && ctx->opt_ucr_loops) {
            out.rws = rws;
            out.tiles = d_tiles;
            out.n_tiles = n_tiles;
            out.windows = pl.windows;
            out.ucr_tables_ready = false;
        }
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_staged, stream));  // every copy out of h_pinned has been enqueued before this point
    ctx->staged_pending = true;
    if (ctx->opt_timing) {
        if (dtw) HIPCHK(ctx, hipEventRecord(ctx->ev1, stream));  // (ED: ev1 was recorded right after the ED kernel)
        ctx->timing_pending = true;  // read by finish_timing() after the call's synchronisation / in kvm_last_kernel_ns
    }
    return KVM_OK;
}

// DTW: the candidates the DP kernel could not decide within rounding (counters[KVM_CNT_AMBIG] of them, in ctx->b_cands2)
// go through the sequential reference-ordered path (kvm_dtw_ref.hip), in batches that fit a bounded slot buffer (2m
// doubles per candidate), each followed by a DP run that reads the exact cb[] arrays.  Answers are appended through the
// same counters.  Called after the main pipeline's counters have reached the host (one extra synchronisation, only when
// such candidates exist: never on continuous data, routinely on integer-valued data and for eps = 0).
static int dtw_resolve_ambiguous(kvm_ctx *ctx, const kvm_query_dev &qd, const kvm_dtw_aux &ax, int64_t n_ambig, verify_out &out,
                                 hipStream_t stream) {
    if (n_ambig <= 0) return KVM_OK;
    const size_t per = (size_t)2 * qd.m * sizeof(double);
    // (128 MiB of cb slots per batch: 2 048 candidates at m = 4096 -- a round of the chains' wave slots; the buffer is grow-only
    //  and stays with the context, so a tie-heavy query must not leave a gigabyte behind)
    int64_t batch = std::max<int64_t>(1, (int64_t)(((size_t)128 << 20) / per));
    if (batch > n_ambig) batch = n_ambig;
    int rc = ensure(ctx, ctx->b_aux2, (size_t)batch * per + 64);
    if (rc) return rc;
    unsigned long long *counters = (unsigned long long *)ctx->b_counters.p;
    kvm_cand *cands = (kvm_cand *)ctx->b_cands2.p;
    double *slots = (double *)ctx->b_aux2.p;
    kvm_dtw_aux ax2 = ax;
    ax2.cb_exact = slots;
    ax2.ambig = nullptr;
    if (kvm_trace_on())
        fprintf(stderr, "[kvm] dtw sequential path: %lld candidates, m %d rho %d, batches of %lld\n", (long long)n_ambig, qd.m,
                qd.rho, (long long)batch);
    // KVM_OPT_UCR_LOOPS (UcrDtwQueryExecutor's never-cleared cb1 / cb2, kvm_ucr.hip): exit depths of every window of the scan
    // once per query, then per candidate the terms earlier windows left behind
    const bool stale = out.tiles != nullptr;
    int32_t *depth1 = nullptr, *depth2 = nullptr, *tmax1 = nullptr, *tmax2 = nullptr;
    if (stale) {
        if ((rc = ensure(ctx, ctx->b_ucr, ((size_t)2 * out.windows + 2 * out.n_tiles) * sizeof(int32_t) + 64))) return rc;
        depth1 = (int32_t *)ctx->b_ucr.p;
        depth2 = depth1 + out.windows;
        tmax1 = depth2 + out.windows;
        tmax2 = tmax1 + out.n_tiles;
    }
    for (int64_t off = 0; off < n_ambig; off += batch) {
        const int64_t nb = std::min(batch, n_ambig - off);
        if (stale) {
            HIPCHK(ctx, kvm_launch_ucr_dtw_stale(ctx->d_series, qd, ax, out.tiles, out.n_tiles, out.rws, depth1, depth2, tmax1, tmax2,
                                                 cands + off, nb, slots, out.ucr_tables_ready, stream));
            out.ucr_tables_ready = true;
            KVM_DBG_SYNC(ctx, stream, "ucr_dtw_stale");
        }
        HIPCHK(ctx, kvm_launch_dtw_ref_chains(ctx->d_series, qd, ax, cands + off, nb, slots, stream, stale));
        KVM_DBG_SYNC(ctx, stream, "dtw_ref_chains");
        HIPCHK(ctx, kvm_launch_set_u64(counters + KVM_CNT_CANDS2, (unsigned long long)nb, counters + KVM_CNT_QUEUE, 0ULL, stream));
        HIPCHK(ctx, kvm_launch_dtw_dp(ctx->d_series, ctx->n, qd, ax2, cands + off, counters + KVM_CNT_CANDS2, nb, nullptr,
                                      out.d_off, out.d_dist, out.cap, counters, ctx->num_cu, (int)ctx->opt_dp_kernel,
                                      stream));
        KVM_DBG_SYNC(ctx, stream, "dtw_dp (exact cb)");
    }
    return KVM_OK;
}

static int verify_host(kvm_ctx *ctx, int variant, const double *query, int32_t m, const int32_t *left,
                       const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho, double alpha,
                       double beta, kvm_result **result) {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!result) return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "null result pointer");
    *result = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto tp_ = t0;
    plan pl;
    int rc;
    verify_out out{nullptr, nullptr, 0, true};
    int64_t kernel_ns = 0;
    kvm_query_dev dtw_qd;
    kvm_dtw_aux dtw_ax;
    rc = run_variant(ctx, variant, query, m, left, right, k, shift, epsilon, rho, alpha, beta, pl, out, ctx->stream,
                     &kernel_ns, dtw_qd, dtw_ax);
    if (rc) return rc;
    const int64_t worst = out.cap;
    const bool is_dtw = variant == V_RSM_DTW || variant == V_CNSM_DTW;
    int64_t n_sequential = 0;
    KVM_TP("run_variant (launch)");
    // One synchronisation in the common case: the counters and the first KVM_SPEC_ANSWERS answers are copied back
    // together (answers are rare); a second copy only when there are more.
    static_assert(KVM_CNT_TOTAL <= 256 && (KVM_SPEC_ANSWERS % 2) == 0, "readback kernel layout");
    unsigned long long hc[KVM_CNT_TOTAL] = {0};
    const int64_t spec = std::min<int64_t>(KVM_SPEC_ANSWERS, worst);
    std::vector<int32_t> off;
    std::vector<double> dist;
    for (int pass = 0; pass < 2 && pl.n_tiles + pl.n_tiles_b + pl.n_tiles_c > 0; pass++) {
        if (!ctx->h_readback)
            HIPCHK(ctx, hipHostMalloc(&ctx->h_readback, KVM_CNT_TOTAL * 8 + KVM_SPEC_ANSWERS * 12 + 64, hipHostMallocDefault));
        HIPCHK(ctx, kvm_launch_readback((const unsigned long long *)ctx->b_counters.p, out.d_off, out.d_dist, spec,
                                        ctx->h_readback, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(hc, ctx->h_readback, sizeof hc);
        if (pass == 0 && is_dtw && hc[KVM_CNT_AMBIG] > 0) {
            // candidates within rounding of a tie with eps^2: sequential reference-ordered path, then read back again
            n_sequential = (int64_t)hc[KVM_CNT_AMBIG];
            if ((rc = dtw_resolve_ambiguous(ctx, dtw_qd, dtw_ax, n_sequential, out, ctx->stream))) return rc;
            continue;
        }
        const int64_t got = std::min<int64_t>((int64_t)hc[KVM_CNT_ANSWERS], spec);
        const int32_t *hoff = (const int32_t *)((const unsigned long long *)ctx->h_readback + KVM_CNT_TOTAL);
        const double *hdist = (const double *)(hoff + ((spec + 1) & ~(int64_t)1));
        off.assign(hoff, hoff + got);
        dist.assign(hdist, hdist + got);
        break;
    }
    KVM_TP("sync + counters D2H");
    if (kvm_trace_on())
        fprintf(stderr, "[kvm] tiles %lld+%lld+%lld  retiled A %llu B %llu C %llu  pruner survivors %llu\n", (long long)pl.n_tiles,
                (long long)pl.n_tiles_b, (long long)pl.n_tiles_c, hc[KVM_CNT_TILES_A], hc[KVM_CNT_TILES_B], hc[KVM_CNT_TILES_C],
                hc[KVM_CNT_CANDS1]);
    const int64_t cnt = (int64_t)hc[KVM_CNT_ANSWERS];
    if (cnt > worst) return fail(ctx, KVM_ERR_DEVICE, "answer buffer overflow (internal error)");
    std::unique_ptr<kvm_result> r_owner(new kvm_result());  // (released into *result at the end; freed on every error return)
    kvm_result *r = r_owner.get();
    off.resize((size_t)cnt);
    dist.resize((size_t)cnt);
    if (cnt > spec) {
        // more answers than the readback kernel carries: the rest through a pinned buffer (grow-only)
        const size_t rest = (size_t)(cnt - spec);
        if (ctx->h_readback2_cap < rest * 12 + 64) {
            if (ctx->h_readback2) HIPCHK(ctx, hipHostFree(ctx->h_readback2));
            ctx->h_readback2 = nullptr;
            ctx->h_readback2_cap = 0;
            const size_t want = std::max<size_t>(rest * 12 + 64, (size_t)1 << 20);
            HIPCHK(ctx, hipHostMalloc(&ctx->h_readback2, want, hipHostMallocDefault));
            ctx->h_readback2_cap = want;
        }
        double *pd = (double *)ctx->h_readback2;
        int32_t *po = (int32_t *)(pd + rest);
        HIPCHK(ctx, hipMemcpyAsync(pd, out.d_dist + spec, rest * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(po, out.d_off + spec, rest * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(dist.data() + spec, pd, rest * 8);
        std::memcpy(off.data() + spec, po, rest * 4);
    }
    KVM_TP("answers D2H");
    // answers.sort(Comparator.comparing(Pair::getSecond)) -- stable on an offset-ascending list
    // (K/QueryEngine.java:373) == sort by (distance, offset).
    // Packed records; above a few thousand answers a stable LSD radix sort (offset digits first, then the bit pattern
    // of the non-negative distance, 11 bits per pass, passes whose digit is the same for every record skipped) instead
    // of a comparison sort: 0.46 -> ~0.1 ms for 10^4 answers.
    struct ans_rec {
        double d;
        int32_t o;
    };
    std::vector<ans_rec> recs((size_t)cnt);
    for (int64_t i = 0; i < cnt; i++) recs[(size_t)i] = ans_rec{dist[(size_t)i], off[(size_t)i]};
    if (cnt < 2048) {
        std::sort(recs.begin(), recs.end(), [](const ans_rec &a, const ans_rec &b) { return a.d != b.d ? a.d < b.d : a.o < b.o; });
    } else {
        std::vector<ans_rec> tmp((size_t)cnt);
        ans_rec *src = recs.data(), *dst = tmp.data();
        auto key_of = [](const ans_rec &r, int pass) -> uint32_t {
            if (pass < 3) return ((uint32_t)r.o >> (11 * pass)) & 2047u;  // offsets are positive int32
            uint64_t u;
            std::memcpy(&u, &r.d, 8);  // distances are >= 0 and never NaN: the bit pattern is monotone
            return (uint32_t)(u >> (11 * (pass - 3))) & 2047u;
        };
        for (int pass = 0; pass < 9; pass++) {
            size_t hist[2049] = {0};
            for (int64_t i = 0; i < cnt; i++) hist[key_of(src[i], pass) + 1]++;
            bool uniform = false;
            for (int b = 1; b <= 2048; b++) uniform |= hist[b] == (size_t)cnt;
            if (uniform) continue;
            for (int b = 0; b < 2048; b++) hist[b + 1] += hist[b];
            for (int64_t i = 0; i < cnt; i++) dst[hist[key_of(src[i], pass)]++] = src[i];
            std::swap(src, dst);
        }
        if (src != recs.data()) std::memcpy(recs.data(), src, (size_t)cnt * sizeof(ans_rec));
    }
    r->off.resize((size_t)cnt);
    r->dist.resize((size_t)cnt);
    for (int64_t i = 0; i < cnt; i++) {
        r->off[(size_t)i] = recs[(size_t)i].o;
        r->dist[(size_t)i] = recs[(size_t)i].d;
    }
    KVM_TP("sort answers");
    r->candidates = pl.cnt_candidate;
    r->stats[KVM_STAT_WINDOWS] = pl.windows;
    const bool cnsm = variant == V_CNSM_ED || variant == V_CNSM_DTW;
    const bool dtw = variant == V_RSM_DTW || variant == V_CNSM_DTW;
    r->stats[KVM_STAT_PASS_FILTER] = cnsm ? (int64_t)hc[KVM_CNT_PASS_FILTER] : pl.windows;
    r->stats[KVM_STAT_RECHECKED] = (int64_t)hc[KVM_CNT_RECHECKED];
    r->stats[KVM_STAT_DTW_SEQUENTIAL] = n_sequential;
    if (dtw) {
        r->stats[KVM_STAT_PASS_LB_KIM] = (int64_t)hc[KVM_CNT_PASS_KIM];
        r->stats[KVM_STAT_PASS_LB_KEOGH] = ctx->opt_lb_cascade ? (int64_t)hc[KVM_CNT_PASS_KEOGH] : (int64_t)hc[KVM_CNT_PASS_KIM];
        r->stats[KVM_STAT_PASS_LB_KEOGH2] = (int64_t)hc[KVM_CNT_PASS_KEOGH2];  // DP runs started
    }
    finish_timing(ctx);
    kernel_ns = ctx->last_kernel_ns;
    r->stats[KVM_STAT_KERNEL_NS] = kernel_ns;
    auto t1 = std::chrono::steady_clock::now();
    r->elapsed_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    *result = r_owner.release();
    return KVM_OK;
}

extern "C" int kvm_verify_rsm_ed(kvm_ctx *ctx, const double *query, int32_t m, const int32_t *left,
                                 const int32_t *right, int32_t k, int32_t shift, double epsilon,
                                 kvm_result **result) {
    return kvm_guard<int>(ctx, [&]() -> int {
    return verify_host(ctx, V_RSM_ED, query, m, left, right, k, shift, epsilon, 0, 1.0, 0.0, result);
    });
}
extern "C" int kvm_verify_cnsm_ed(kvm_ctx *ctx, const double *query, int32_t m, const int32_t *left,
                                  const int32_t *right, int32_t k, int32_t shift, double epsilon, double alpha,
                                  double beta, kvm_result **result) {
    return kvm_guard<int>(ctx, [&]() -> int {
    return verify_host(ctx, V_CNSM_ED, query, m, left, right, k, shift, epsilon, 0, alpha, beta, result);
    });
}
extern "C" int kvm_verify_rsm_dtw(kvm_ctx *ctx, const double *query, int32_t m, const int32_t *left,
                                  const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho,
                                  kvm_result **result) {
    return kvm_guard<int>(ctx, [&]() -> int {
    return verify_host(ctx, V_RSM_DTW, query, m, left, right, k, shift, epsilon, rho, 1.0, 0.0, result);
    });
}
extern "C" int kvm_verify_cnsm_dtw(kvm_ctx *ctx, const double *query, int32_t m, const int32_t *left,
                                   const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho,
                                   double alpha, double beta, kvm_result **result) {
    return kvm_guard<int>(ctx, [&]() -> int {
    return verify_host(ctx, V_CNSM_DTW, query, m, left, right, k, shift, epsilon, rho, alpha, beta, result);
    });
}

extern "C" int kvm_verify_device(kvm_ctx *ctx, int32_t variant, const double *query, int32_t m, const int32_t *left,
                                 const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho,
                                 double alpha, double beta, int32_t *d_out_off, double *d_out_dist, int64_t cap,
                                 int64_t *d_out_count, int64_t *cnt_candidate, void *stream) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (variant < 0 || variant > 3 || !d_out_off || !d_out_dist || !d_out_count || cap < 0)
        return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "bad kvm_verify_device arguments");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    plan pl;
    verify_out out{d_out_off, d_out_dist, cap, false};
    kvm_query_dev dtw_qd;
    kvm_dtw_aux dtw_ax;
    int rc = run_variant(ctx, variant, query, m, left, right, k, shift, epsilon, rho, alpha, beta, pl, out, s, nullptr, dtw_qd,
                         dtw_ax);
    if (rc) return rc;
    if (cnt_candidate) *cnt_candidate = pl.cnt_candidate;
    if (variant >= V_RSM_DTW && pl.n_tiles + pl.n_tiles_b + pl.n_tiles_c > 0) {
        // DTW: the number of candidates that need the sequential reference-ordered path is only known on the device;
        // this one small read is the only point where a kvm_verify_device call waits for the stream
        unsigned long long n_ambig = 0;
        HIPCHK(ctx, hipMemcpyAsync(&n_ambig, (unsigned long long *)ctx->b_counters.p + KVM_CNT_AMBIG, sizeof n_ambig,
                                   hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        if (n_ambig > 0 && (rc = dtw_resolve_ambiguous(ctx, dtw_qd, dtw_ax, (int64_t)n_ambig, out, s))) return rc;
    }
    if (pl.n_tiles + pl.n_tiles_b + pl.n_tiles_c == 0) {
        HIPCHK(ctx, hipMemsetAsync(d_out_count, 0, sizeof(int64_t), s));
    } else {
        HIPCHK(ctx, hipMemcpyAsync(d_out_count, (unsigned long long *)ctx->b_counters.p + KVM_CNT_ANSWERS,
                                   sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    }
    // Everything this call has enqueued -- the pipeline, the sequential tie path's kernels, the copy of the answer count out
    // of the counters -- reads the context's device scratch: the next call, possibly on another stream, must be ordered
    // behind all of it before it reuses the buffers.  That is a device-side wait on ev_done (run_variant); the HOST only
    // ever waits for ev_staged, recorded behind the last read of the pinned staging buffer, so it can prepare the next
    // call while this one still runs.
    HIPCHK(ctx, hipEventRecord(ctx->ev_done, s));
    ctx->done_stream = s;
    ctx->done_pending = true;
    return KVM_OK;
    });
}

extern "C" int kvm_pack_answers_device(kvm_ctx *ctx, const int32_t *d_off, const double *d_dist, const int64_t *d_count,
                                       int64_t pack_cap, double *d_packed, void *stream) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!d_off || !d_dist || !d_count || !d_packed || pack_cap < 0)
        return fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "bad kvm_pack_answers_device arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, kvm_launch_pack_answers(d_off, d_dist, d_count, pack_cap, (double2 *)d_packed,
                                        stream ? (hipStream_t)stream : ctx->stream));
    return KVM_OK;
    });
}

// ---- results ----------------------------------------------------------------------------------------------
extern "C" int64_t kvm_result_count(const kvm_result *r) { return r ? (int64_t)r->off.size() : 0; }
extern "C" int64_t kvm_result_candidates(const kvm_result *r) { return r ? r->candidates : 0; }
extern "C" double kvm_result_elapsed_ms(const kvm_result *r) { return r ? r->elapsed_ms : 0.0; }
extern "C" const int32_t *kvm_result_offsets(const kvm_result *r) { return r ? r->off.data() : nullptr; }
extern "C" const double *kvm_result_distances(const kvm_result *r) { return r ? r->dist.data() : nullptr; }
extern "C" int64_t kvm_result_stat(const kvm_result *r, int32_t which) {
    return (r && which >= 0 && which < KVM_STAT_COUNT) ? r->stats[which] : 0;
}
extern "C" void kvm_result_free(kvm_result *r) { delete r; }
