// kvm_stream.cpp -- host layer: several phase-2 calls in flight, one host thread per context (kvm_run_jobs).
// The reference's engines are single-threaded and not re-entrant (K/QueryEngine.java:61-68); a host that wants several
// queries in flight runs one engine instance per thread.  This is that loop in native code, so that a driver written in a
// language with a global interpreter lock (bench.py, the tests) measures the library and not its own lock: the Python
// threads of kvmatch.QueryStream peak at 80 us per call of an early-abandoning RSM-ED stream, these threads at 54 - 60
// (profiles/r04_stream_native_0.log).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <pthread.h>

#include <string>

#include "../../include/kvmatch.h"

struct kvm_ctx;
void kvm_set_error_noexcept(kvm_ctx *ctx, const char *msg) noexcept;  // kvm_api.hip

#include "kvm_host_threads.h"
using kvm_host::thread_group;
using kvm_host::first_call_alone;
using kvm_host::in_flight_mark;

static int run_one_call(kvm_ctx *c, const kvm_job &j, kvm_result **r) {
    switch (j.variant) {
        case 0: return kvm_verify_rsm_ed(c, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, r);
        case 1: return kvm_verify_cnsm_ed(c, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, j.alpha, j.beta, r);
        case 2: return kvm_verify_rsm_dtw(c, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, j.rho, r);
        case 3: return kvm_verify_cnsm_dtw(c, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, j.rho, j.alpha, j.beta, r);
    }
    return KVM_ERR_ILLEGAL_ARGUMENT;
}
static int run_one(kvm_ctx *c, const kvm_job &j, kvm_result **r) {
    first_call_alone guard;
    return run_one_call(c, j, r);
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
        // A failing job's message lives in the context that ran it (a context's text is not cleared by later successes, and
        // another context may still carry the text of a failure handled long ago): every thread keeps the message of ITS
        // lowest-index failure, and the one of the call's first failing job is copied to ctxs[0] below -- that is where
        // a caller reads kvm_last_error after kvm_run_jobs.
        std::vector<int64_t> bad_job((size_t)n_ctx, -1);
        std::vector<std::string> bad_msg((size_t)n_ctx);
        auto loop = [&](int32_t t) {
            kvm_ctx *c = ctxs[t];
            for (;;) {
                if (failed.load(std::memory_order_relaxed)) return;
                const int64_t i = next.fetch_add(1);
                if (i >= n_jobs) return;
                const int rc = run_one(c, jobs[i], &results[i]);
                rcs[(size_t)i] = rc;
                if (rc != KVM_OK) {
                    failed.store(true);
                    if (bad_job[(size_t)t] < 0) {
                        bad_job[(size_t)t] = i;
                        try {
                            bad_msg[(size_t)t] = kvm_last_error(c);
                        } catch (...) {  // (no memory for the text: the status has to do)
                        }
                    }
                }
            }
        };
        const int32_t nt = (int32_t)(n_jobs < n_ctx ? (n_jobs > 0 ? n_jobs : 1) : n_ctx);
        {
            auto get = [&](int32_t t) { return ctxs[t]; };
            in_flight_mark<decltype(get)> mark(get, n_ctx, nt);
            thread_group g;
            try {
                for (int32_t t = 1; t < nt; t++) g.start([&loop, t] { loop(t); });
            } catch (...) {
                failed.store(true);  // the threads that did start take no further job; ~thread_group joins them
                throw;
            }
            loop(0);  // the calling thread drives the first context
        }
        int first = KVM_OK;
        int64_t first_job = -1;
        for (int64_t i = 0; i < n_jobs; i++) {
            if (status) status[i] = rcs[(size_t)i];
            if (first == KVM_OK && rcs[(size_t)i] != KVM_OK) {
                first = rcs[(size_t)i];
                first_job = i;
            }
        }
        for (int32_t t = 1; t < n_ctx && first_job >= 0; t++)   // (t = 0: the text is where it belongs already)
            if (bad_job[(size_t)t] == first_job) kvm_set_error_noexcept(ctxs[0], bad_msg[(size_t)t].c_str());
        return first;
    } catch (const std::bad_alloc &) {
        kvm_set_error_noexcept(ctxs[0], "kvm_run_jobs: out of host memory (std::bad_alloc)");
        return KVM_ERR_NO_MEMORY;
    } catch (const std::exception &e) {  // (std::system_error from a thread that could not be started)
        kvm_set_error_noexcept(ctxs[0], e.what());
        return KVM_ERR_DEVICE;
    } catch (...) {  // nothing foreign crosses the C ABI (kvm_guard.h)
        kvm_set_error_noexcept(ctxs[0], "kvm_run_jobs: unknown C++ exception");
        return KVM_ERR_DEVICE;
    }
}

// ---- the device-resident form: slots enqueue, the caller's own thread runs the collectives (kvmatch.distributed.ShardedStream)
int kvm_poison_packed_device(kvm_ctx *ctx, double *d_packed, void *stream);  // kvm_verify.hip (internal)
int kvm_record_event_device(kvm_ctx *ctx, void *event, void *stream);         // kvm_verify.hip (internal)

extern "C" int kvm_run_device_jobs(const kvm_device_slot *slots, int32_t n_slots, const kvm_job *jobs, int64_t n_jobs,
                                   volatile int32_t *produced, const volatile int32_t *consumed, const volatile int32_t *stop,
                                   int64_t *cnt_candidate, int32_t *status) {
    if (!slots || n_slots < 1 || n_jobs < 0 || (n_jobs > 0 && (!jobs || !produced || !consumed || !stop || !status)))
        return KVM_ERR_ILLEGAL_ARGUMENT;
    for (int32_t t = 0; t < n_slots; t++)
        if (!slots[t].ctx) return KVM_ERR_INVALID_HANDLE;
    try {
        std::atomic<int> first_rc{KVM_OK};
        std::atomic<bool> abort{false};  // this call's own stop (a helper thread could not be started)
        // (the flags are shared with a caller written in another language: plain int32 words, read and written here with
        // acquire / release so that a job's status and enqueued work are visible before its `produced` flag)
        auto flag = [](const volatile int32_t *p) { return __atomic_load_n(const_cast<const int32_t *>(p), __ATOMIC_ACQUIRE) != 0; };
        auto loop = [&](int32_t t) {
            const kvm_device_slot &s = slots[t];
            bool failed = false;  // a slot that has failed keeps taking part with the failure mark (every rank ends the stream there)
            for (int64_t i = t; i < n_jobs; i += n_slots) {
                // (a slot waits for the caller's exchange of its previous job: milliseconds in a stream that fills the GPU -- the
                // poll backs off to 20 us so that the waiting slots of a rank do not keep a core busy each)
                for (int nap = 5; i - n_slots >= 0 && !flag(&consumed[i - n_slots]) && !flag(stop) && !abort.load(std::memory_order_relaxed);
                     nap = nap < 20 ? nap * 2 : 20)
                    std::this_thread::sleep_for(std::chrono::microseconds(nap));
                if (flag(stop) || abort.load(std::memory_order_relaxed)) return;
                int rc = KVM_OK;
                if (!failed) {
                    const kvm_job &j = jobs[i];
                    int64_t cnt = 0;
                    first_call_alone guard;
                    rc = kvm_verify_device(s.ctx, j.variant, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, j.rho, j.alpha, j.beta,
                                           s.d_off, s.d_dist, s.cap, s.d_count, &cnt, s.stream);
                    if (rc == KVM_OK) rc = kvm_pack_answers_device(s.ctx, s.d_off, s.d_dist, s.d_count, s.pack_cap, s.d_packed, s.stream);
                    if (cnt_candidate) cnt_candidate[i] = cnt;
                    if (rc != KVM_OK) failed = true;
                }
                if (failed) {
                    if (rc == KVM_OK) rc = KVM_ERR_DEVICE;   // (a job behind the slot's failure)
                    (void)kvm_poison_packed_device(s.ctx, s.d_packed, s.stream);
                    int expected = KVM_OK;
                    first_rc.compare_exchange_strong(expected, rc);
                }
                if (s.ready && kvm_record_event_device(s.ctx, s.ready, s.stream) != KVM_OK && rc == KVM_OK) {
                    rc = KVM_ERR_DEVICE;   // (the caller would wait for an event that was never recorded: end the stream here)
                    failed = true;
                    int expected = KVM_OK;
                    first_rc.compare_exchange_strong(expected, rc);
                }
                status[i] = rc;
                __atomic_store_n(const_cast<int32_t *>(&produced[i]), 1, __ATOMIC_RELEASE);
            }
        };
        {
            auto get = [&](int32_t t) { return slots[t].ctx; };
            in_flight_mark<decltype(get)> mark(get, n_slots, (int32_t)std::min<int64_t>(n_slots, std::max<int64_t>(n_jobs, 1)));
            thread_group g;
            try {
                for (int32_t t = 1; t < n_slots; t++) g.start([&loop, t] { loop(t); });
            } catch (...) {
                abort.store(true);
                throw;
            }
            loop(0);
        }
        return first_rc.load();
    } catch (const std::bad_alloc &) {
        return KVM_ERR_NO_MEMORY;
    } catch (const std::exception &) {
        return KVM_ERR_DEVICE;
    } catch (...) {  // nothing foreign crosses the C ABI (kvm_guard.h)
        return KVM_ERR_DEVICE;
    }
}
