// kvm_api.hip -- the C ABI of include/kvmatch.h, core: context life cycle, options, error text, scratch helpers and series
// staging (TimeSeriesOperator side).  The other translation units of the ABI: kvm_api_index.hip (index build, step 1),
// kvm_api_model.hip (time model, rho, the JDK sort order: host only), kvm_plan.hip (intervals -> runs -> tile counts),
// kvm_verify.hip (the phase-2 driver and its entry points), kvm_stream.cpp (several calls in flight), kvm_index_file.cpp.
// Host orchestration only; every distance is computed by the HIP kernels in kvm_ed*.hip / kvm_dtw*.hip.  There is
// deliberately no CPU fallback: every entry point fails with KVM_ERR_DEVICE when HIP is unusable.
#include "kvm_api_internal.h"

static thread_local std::string g_open_err;

// Every event / the stream a context owns, destroyed exactly once (kvm_open's failure paths and kvm_close share this list).
static void destroy_events(kvm_ctx *ctx) {
    hipEvent_t *evs[] = {&ctx->ev0, &ctx->ev1, &ctx->ev_staged, &ctx->ev_done, &ctx->ev_env};
    for (hipEvent_t *e : evs) {
        if (*e) hipEventDestroy(*e);
        *e = nullptr;
    }
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    ctx->stream = nullptr;
}

int kvm_fail(kvm_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg; else g_open_err = msg;
    return code;
}

void kvm_set_error(kvm_ctx *ctx, const std::string &msg) { kvm_fail(ctx, 0, msg); }  // for the host-only translation units
void kvm_set_error_noexcept(kvm_ctx *ctx, const char *msg) noexcept {
    try {
        (ctx ? ctx->err : g_open_err).assign(msg);
    } catch (...) {  // (not even the message fits: the status code has to do)
    }
}


int kvm_ensure(kvm_ctx *ctx, kvm_dev_buf &b, size_t bytes) {
    if (bytes <= b.cap) return KVM_OK;
    if (b.p) HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return KVM_OK;
}

int kvm_ensure_pinned(kvm_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->h_pinned_cap) return KVM_OK;
    if (ctx->h_pinned) HIPCHK(ctx, hipHostFree(ctx->h_pinned));
    ctx->h_pinned = nullptr;
    ctx->h_pinned_cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(ctx, hipHostMalloc(&ctx->h_pinned, want, hipHostMallocDefault));
    ctx->h_pinned_cap = want;
    return KVM_OK;
}

// (0.6: round 6 -- the group entry points, KVM_OPT_ED_FORCE / _ED_LAST_SHAPE / _SERIALIZE_ENQUEUE, kvm_abi_version() == 6)
extern "C" const char *kvm_version(void) { return "kvmatch-hip 0.6 (gfx950)"; }

extern "C" const char *kvm_last_error(const kvm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_open_err.c_str(); }

extern "C" int kvm_open(int device, kvm_ctx **out) {
    return kvm_guard<int>(nullptr, [&]() -> int {
    if (!out) return kvm_fail(nullptr, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_open: null out pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return kvm_fail(nullptr, KVM_ERR_DEVICE,
                    std::string("kvm_open: no usable HIP device (") + hipGetErrorString(e) +
                        "); this library has no CPU fallback");
    if (device < 0 || device >= ndev)
        return kvm_fail(nullptr, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_open: device ordinal out of range");
    kvm_ctx *ctx = new kvm_ctx();
    ctx->device = device;
    static const bool serialize_default = [] {   // (read once per process)
        const char *e = getenv("KVM_SERIALIZE_ENQUEUE");
        return e && e[0] && e[0] != '0';
    }();
    ctx->opt_serialize_enqueue = serialize_default ? 1 : 0;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&ctx->stream)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_staged, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming)) != hipSuccess) {
        std::string msg = std::string("kvm_open: ") + hipGetErrorString(e);
        destroy_events(ctx);
        delete ctx;
        return kvm_fail(nullptr, KVM_ERR_DEVICE, msg);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (std::strstr(prop.gcnArchName, "gfx950") == nullptr) {
            std::string msg = std::string("kvm_open: built for gfx950 only, device is ") + prop.gcnArchName;
            destroy_events(ctx);
            delete ctx;
            return kvm_fail(nullptr, KVM_ERR_DEVICE, msg);
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

// the group layer runs a slot on its context's own stream (kvm_sharded.cpp)
void *kvm_ctx_stream_handle(kvm_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int kvm_close(kvm_ctx *ctx) {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    free_series(ctx);
    kvm_dev_buf *bufs[] = {&ctx->b_query, &ctx->b_tiles, &ctx->b_out_off, &ctx->b_out_dist, &ctx->b_counters,
                           &ctx->b_cands, &ctx->b_cands2, &ctx->b_aux,   &ctx->b_aux2,    &ctx->b_scratch, &ctx->b_plans,
                           &ctx->b_ucr,   &ctx->b_edx};
    for (kvm_dev_buf *b : bufs)
        if (b->p) hipFree(b->p);
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    if (ctx->h_readback) hipHostFree(ctx->h_readback);
    if (ctx->h_readback2) hipHostFree(ctx->h_readback2);
    kvm_idx_block_drop();
    destroy_events(ctx);
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
                return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "KVM_OPT_ED_KERNEL in {0,1,3}");
            ctx->opt_ed_kernel = value;
            return KVM_OK;
        case KVM_OPT_TIMING: ctx->opt_timing = value ? 1 : 0; return KVM_OK;
        case KVM_OPT_DP_KERNEL: ctx->opt_dp_kernel = (value >= 0 && value <= 3) ? value : 0; return KVM_OK;
        case KVM_OPT_SORT_VIOLATION: ctx->opt_sort_violation = value ? 1 : 0; return KVM_OK;
        case KVM_OPT_UCR_LOOPS: ctx->opt_ucr_loops = value ? 1 : 0; return KVM_OK;
        case KVM_OPT_STRICT_BOUNDS: ctx->opt_strict_bounds = value ? 1 : 0; return KVM_OK;
        case KVM_OPT_IN_FLIGHT: ctx->opt_in_flight = value < 1 ? 1 : (value > 64 ? 64 : value); return KVM_OK;
        case KVM_OPT_ED_FORCE: {
            const auto split_ok = [](int64_t f) { return f == 0 || f == 1 || f == 2 || f == 4; };
            if (value < 0 || value >= (1 << 16) || !split_ok(value & 15) || !split_ok((value >> 4) & 15) || ((value >> 8) & 3) == 3 ||
                ((value >> 10) & 3) == 3 || ((value >> 12) & 3) == 3)
                return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "KVM_OPT_ED_FORCE: see KVM_ED_FORCE in kvmatch.h");
            ctx->opt_ed_force = value;
            return KVM_OK;
        }
        case KVM_OPT_SERIALIZE_ENQUEUE: ctx->opt_serialize_enqueue = value ? 1 : 0; return KVM_OK;
    }
    return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "unknown option");
}

// kvm_stream.cpp (internal): the contexts of one kvm_run_jobs / kvm_run_device_jobs call, for its duration (0 afterwards)
void kvm_set_run_in_flight(kvm_ctx *ctx, int n) noexcept {
    if (ctx) ctx->run_in_flight = n;
}

// The events around the timed kernel are read only once everything behind them has been enqueued (and, on the host path,
// after the call's single synchronisation): waiting for them inside the pipeline would put a host wake-up between the
// kernels of a call.
void kvm_finish_timing(kvm_ctx *ctx) {
    if (!ctx->timing_pending) return;
    ctx->timing_pending = false;
    float ms = 0;
    if (hipEventSynchronize(ctx->ev1) == hipSuccess && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess)
        ctx->last_kernel_ns = (int64_t)((double)ms * 1e6);
}
extern "C" int kvm_measure_device(kvm_ctx *ctx, int32_t what, double *value) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!value || what < 0 || what > 2) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_measure_device: what = 0, 1 or 2 and a result pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, kvm_measure_device_rate(what, ctx->num_cu, ctx->stream, ctx->ev0, ctx->ev1, value));
    return KVM_OK;
    });
}

int64_t kvm_last_kernel_ns(const kvm_ctx *ctx) {
    if (!ctx) return 0;
    kvm_finish_timing(const_cast<kvm_ctx *>(ctx));
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
        case KVM_OPT_IN_FLIGHT: return ctx->opt_in_flight;
        case KVM_OPT_ED_FORCE: return ctx->opt_ed_force;
        case KVM_OPT_SERIALIZE_ENQUEUE: return ctx->opt_serialize_enqueue;
        case KVM_OPT_ED_LAST_SHAPE: return ctx->last_ed_shape;
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
        return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "series length must be in [1, 2^31-1]");
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
    if (!host_values) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "null series");
    int rc = alloc_series(ctx, n);
    if (rc) return rc;
    if (hipMemcpyAsync(ctx->d_series, host_values, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        free_series(ctx);
        return kvm_fail(ctx, KVM_ERR_DEVICE, "copying the series to the device failed");
    }
    return finish_load(ctx);
    });
}

extern "C" int kvm_load_series_be64(kvm_ctx *ctx, const void *be_bytes, int64_t n) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!be_bytes) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "null series");
    int rc = alloc_series(ctx, n);
    if (rc) return rc;
    if (hipMemcpyAsync(ctx->d_series, be_bytes, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        kvm_launch_bswap64((uint64_t *)ctx->d_series, n, ctx->stream) != hipSuccess) {
        hipStreamSynchronize(ctx->stream);
        free_series(ctx);
        return kvm_fail(ctx, KVM_ERR_DEVICE, "staging the big-endian series failed");
    }
    return finish_load(ctx);
    });
}

extern "C" int kvm_load_series_file(kvm_ctx *ctx, const char *path, int64_t n) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!path) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "null path");
    if (n < 1 || n > 2147483647LL)  // (before the file is touched: a bad length is the caller's error, not an I/O error)
        return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "series length must be in [1, 2^31-1]");
    int fd = open(path, O_RDONLY);
    if (fd < 0) return kvm_fail(ctx, KVM_ERR_IO, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || (int64_t)st.st_size < n * 8) {
        close(fd);
        return kvm_fail(ctx, KVM_ERR_IO, std::string(path) + ": shorter than n*8 bytes");
    }
    void *map = mmap(nullptr, (size_t)n * 8, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return kvm_fail(ctx, KVM_ERR_IO, std::string("mmap failed for ") + path);
    int rc = kvm_load_series_be64(ctx, map, n);
    munmap(map, (size_t)n * 8);
    return rc;
    });
}

extern "C" int kvm_attach_series_device(kvm_ctx *ctx, const double *device_values, int64_t n) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!device_values || n < 1 || n > 2147483647LL) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "bad device series");
    // device-to-device copy into the ctx's own padded buffer (see alloc_series); the caller's array is not
    // referenced after this call returns
    int rc = alloc_series(ctx, n);
    if (rc) return rc;
    if (hipMemcpyAsync(ctx->d_series, device_values, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
        free_series(ctx);
        return kvm_fail(ctx, KVM_ERR_DEVICE, "copying the device series failed");
    }
    return finish_load(ctx);
    });
}

extern "C" int kvm_share_series(kvm_ctx *ctx, const kvm_ctx *owner) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx || !owner) return KVM_ERR_INVALID_HANDLE;
    if (ctx == owner) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "a context cannot share its own series");
    if (!owner->d_series || !owner->d_prefix) return kvm_fail(ctx, KVM_ERR_NO_SERIES, "the owner context has no series loaded");
    if (owner->device != ctx->device) return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "contexts on different devices cannot share a series");
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
    if (!ctx->d_series) return kvm_fail(ctx, KVM_ERR_NO_SERIES, "no series loaded");
    // K/operator/file/TimeSeriesFileOperator.java:55-57
    if (left < 1 || (int64_t)left + length - 1 > ctx->n || length < 1 || !out) {
        char msg[96];
        snprintf(msg, sizeof msg, "left: %d, length: %d", left, length);
        return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, msg);
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
    if (!ctx->d_series) return kvm_fail(ctx, KVM_ERR_NO_SERIES, "no series loaded");
    if (w < 1 || first < 1 || count < 0 || !out || (int64_t)first + count - 1 + w - 1 > ctx->n)
        return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_window_means: window range outside the series");
    if (count == 0) return KVM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = kvm_ensure(ctx, ctx->b_scratch, (size_t)count * 8);
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

