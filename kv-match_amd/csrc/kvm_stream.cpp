// kvm_stream.cpp -- host layer: several phase-2 calls in flight, one host thread per context (kvm_run_jobs).
// The reference's engines are single-threaded and not re-entrant (K/QueryEngine.java:61-68); a host that wants several
// queries in flight runs one engine instance per thread.  This is that loop in native code, so that a driver written in a
// language with a global interpreter lock (bench.py, the tests) measures the library and not its own lock: the Python
// threads of kvmatch.QueryStream peak at 80 us per call of an early-abandoning RSM-ED stream, these threads at 54 - 60
// (profiles/r04_stream_native_0.log).
#include <atomic>
#include <exception>
#include <new>
#include <thread>
#include <vector>

#include "../../include/kvmatch.h"

static int run_one(kvm_ctx *c, const kvm_job &j, kvm_result **r) {
    switch (j.variant) {
        case 0: return kvm_verify_rsm_ed(c, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, r);
        case 1: return kvm_verify_cnsm_ed(c, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, j.alpha, j.beta, r);
        case 2: return kvm_verify_rsm_dtw(c, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, j.rho, r);
        case 3: return kvm_verify_cnsm_dtw(c, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, j.rho, j.alpha, j.beta, r);
    }
    return KVM_ERR_ILLEGAL_ARGUMENT;
}

extern "C" int kvm_run_jobs(kvm_ctx *const *ctxs, int32_t n_ctx, const kvm_job *jobs, int64_t n_jobs, kvm_result **results,
                            int32_t *status) {
    if (!ctxs || n_ctx < 1 || n_jobs < 0 || (n_jobs > 0 && (!jobs || !results))) return KVM_ERR_ILLEGAL_ARGUMENT;
    for (int32_t t = 0; t < n_ctx; t++)
        if (!ctxs[t]) return KVM_ERR_INVALID_HANDLE;
    try {
        for (int64_t i = 0; i < n_jobs; i++) {
            results[i] = nullptr;
            if (status) status[i] = KVM_OK;
        }
        std::atomic<int64_t> next{0};
        std::atomic<bool> failed{false};
        std::vector<int> rcs((size_t)n_jobs, KVM_OK);
        auto loop = [&](kvm_ctx *c) {
            for (;;) {
                if (failed.load(std::memory_order_relaxed)) return;
                const int64_t i = next.fetch_add(1);
                if (i >= n_jobs) return;
                const int rc = run_one(c, jobs[i], &results[i]);
                rcs[(size_t)i] = rc;
                if (rc != KVM_OK) failed.store(true);
            }
        };
        const int32_t nt = (int32_t)(n_jobs < n_ctx ? (n_jobs > 0 ? n_jobs : 1) : n_ctx);
        std::vector<std::thread> th;
        th.reserve((size_t)nt);
        for (int32_t t = 1; t < nt; t++) th.emplace_back(loop, ctxs[t]);
        loop(ctxs[0]);  // the calling thread drives the first context
        for (auto &x : th) x.join();
        int first = KVM_OK;
        for (int64_t i = 0; i < n_jobs; i++) {
            if (status) status[i] = rcs[(size_t)i];
            if (first == KVM_OK && rcs[(size_t)i] != KVM_OK) first = rcs[(size_t)i];
        }
        return first;
    } catch (const std::bad_alloc &) {
        return KVM_ERR_NO_MEMORY;
    } catch (const std::exception &) {  // (std::system_error from a thread that could not be started)
        return KVM_ERR_DEVICE;
    }
}
