// kvm_api_internal.h -- what the translation units of the C ABI share (kvm_api*.hip, kvm_plan.hip, kvm_verify.hip): the error
// and scratch helpers, the trace switches, and the plan a call is driven by.  Host only.
#pragma once
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

#define KVM_MAX_SCAN_DATA_LENGTH 40000  // K/NormQueryEngine.java:60, K/NormQueryEngineDtw.java:61
#define KVM_TILE_MAX 64                 // candidates per tile: wave ED kernel, DTW filter (one lane each)
#define KVM_TILE_TILED 32               // candidates per tile of the LDS-tiled ED kernel (== ED_ROW_CAND)

// kvm_api.hip
int kvm_fail(kvm_ctx *ctx, int code, const std::string &msg);   // stores the message, returns `code`
int kvm_ensure(kvm_ctx *ctx, kvm_dev_buf &b, size_t bytes);     // grow-only device scratch
int kvm_ensure_pinned(kvm_ctx *ctx, size_t bytes);              // grow-only pinned staging block
void kvm_finish_timing(kvm_ctx *ctx);
#define HIPCHK(ctx, call)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return kvm_fail(ctx, KVM_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
// kvm_api_index.hip
void kvm_idx_block_drop();  // pinned result block of the index build, kept per process

// kvm_plan.hip
bool kvm_trace_on();       // KVM_TRACE=1: per-stage host timings of every call on stderr
bool kvm_debug_sync_on();  // KVM_DEBUG_SYNC=1: wait for the stream after every stage of a DTW call and name it on stderr
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

int kvm_build_plan(kvm_ctx *ctx, bool cnsm, int32_t m, const int32_t *left, const int32_t *right, int32_t k,
                      int32_t shift, int32_t tile_max, bool hybrid, plan &pl, int32_t rho_strict = -1);
void kvm_query_envelope(const std::vector<double> &q, int r, std::vector<double> &u, std::vector<double> &l);
