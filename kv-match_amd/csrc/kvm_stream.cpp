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

// The helper threads of a call are PERSISTENT (round 5): a process-wide pool of workers that sleep between calls, grown on
// demand and never shrunk.  Rounds 4's form started n_ctx - 1 fresh std::threads per kvm_run_jobs call and joined them at
// its end; every such thread initialises its own HIP thread state on its first launch and tears it down when it exits, and
// a tool that hooks the runtime (rocprofv3's kernel tracing) registers / drops per-thread buffers at the same two points --
// hundreds of times in a bench run.  With workers that live as long as the process, that happens once per worker.
// A call hands its loops to the pool and waits for all of them on every path out (thread_group's destructor): the loops
// reference the call's stack frame.  If a worker cannot be started (std::system_error) the loops already handed over are
// told to stop by the caller's flag and waited for; the C ABI hands back a status.
namespace {
class worker_pool {
public:
    struct ticket {   // one per call: how many of its loops are still running
        std::mutex mu;
        std::condition_variable cv;
        int pending = 0;
    };
    // runs fn on a pool thread; throws (std::system_error / std::bad_alloc) if no worker can be had
    void submit(ticket &t, std::function<void()> fn) {
        std::unique_lock<std::mutex> lk(mu_);
        // every queued task needs a sleeping worker of its own (a worker that has been notified but has not woken yet still
        // counts as sleeping, and its task is still queued); busy workers are not counted on
        if ((int)q_.size() + 1 > waiting_) {
            std::thread th([this] { work(); });   // may throw: nothing has been queued yet
            th.detach();                          // (lives until the process ends; it only ever blocks on cv_)
        }
        q_.push_back(task{&t, std::move(fn)});    // (may throw: a worker started above just goes to sleep)
        {
            std::lock_guard<std::mutex> tl(t.mu);
            t.pending++;
        }
        lk.unlock();
        cv_.notify_one();
    }
    static void wait(ticket &t) {
        std::unique_lock<std::mutex> lk(t.mu);
        t.cv.wait(lk, [&] { return t.pending == 0; });
    }
    static worker_pool &get() {
        // never destroyed: its threads may outlive static destruction.  A fork()ed child inherits the object but none of
        // its threads (and possibly a locked mutex): it starts over with an empty pool of its own.
        static std::once_flag once;
        std::call_once(once, [] {
            instance() = new worker_pool();
            pthread_atfork(nullptr, nullptr, [] { instance() = new worker_pool(); });
        });
        return *instance();
    }

private:
    static worker_pool *&instance() {
        static worker_pool *p = nullptr;
        return p;
    }
    struct task {
        ticket *t;
        std::function<void()> fn;
    };
    void work() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            if (q_.empty()) {
                waiting_++;
                cv_.wait(lk, [&] { return !q_.empty(); });
                waiting_--;
            }
            task k = std::move(q_.front());
            q_.pop_front();
            lk.unlock();
            try {
                k.fn();
            } catch (...) {  // (the loops catch their own; nothing may escape a detached thread)
            }
            {
                std::lock_guard<std::mutex> tl(k.t->mu);
                k.t->pending--;
                // notify under the lock: the waiter owns the ticket and may destroy it as soon as it sees pending == 0
                k.t->cv.notify_all();
            }
            lk.lock();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<task> q_;
    int waiting_ = 0;   // workers asleep in cv_.wait
};
}  // namespace

struct thread_group {
    worker_pool::ticket t;
    void start(std::function<void()> fn) { worker_pool::get().submit(t, std::move(fn)); }
    ~thread_group() { worker_pool::wait(t); }
};

static int run_one_call(kvm_ctx *c, const kvm_job &j, kvm_result **r) {
    switch (j.variant) {
        case 0: return kvm_verify_rsm_ed(c, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, r);
        case 1: return kvm_verify_cnsm_ed(c, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, j.alpha, j.beta, r);
        case 2: return kvm_verify_rsm_dtw(c, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, j.rho, r);
        case 3: return kvm_verify_cnsm_dtw(c, j.query, j.m, j.left, j.right, j.k, j.shift, j.epsilon, j.rho, j.alpha, j.beta, r);
    }
    return KVM_ERR_ILLEGAL_ARGUMENT;
}
// A thread's FIRST call runs alone: it is where the HIP runtime (and a tool that hooks it) sets up its per-thread state, and
// eight threads doing that at the same moment is what `rocprofv3 --kernel-trace -- python3 bench.py` fell over in (SIGSEGV in a
// copy inside the tool's launch callback, seconds into the run, two attempts of three with the pool and without; with this
// one of three: profiles/r05_prof_native_crash.log; what ended it is kvm_enqueue_lock in kvm_verify.hip).  Once per thread of the
// process; the calls after it run freely.
struct first_call_alone {
    std::unique_lock<std::mutex> lk;
    first_call_alone() {
        static std::mutex mu;
        if (!done()) lk = std::unique_lock<std::mutex>(mu);
    }
    ~first_call_alone() { done() = true; }
    static bool &done() {
        thread_local bool d = false;
        return d;
    }
};
// The launcher of the ED pruner plans a list differently when launches of other contexts run next to its own (KVM_OPT_IN_FLIGHT):
// the contexts of a call here are told for its duration.
void kvm_set_run_in_flight(kvm_ctx *ctx, int n) noexcept;  // kvm_api.hip (internal)
template <class Get>
struct in_flight_mark {
    Get get;
    int32_t n;
    in_flight_mark(Get g, int32_t count, int32_t busy) : get(g), n(count) {
        for (int32_t t = 0; t < n; t++) kvm_set_run_in_flight(get(t), busy);
    }
    ~in_flight_mark() {
        for (int32_t t = 0; t < n; t++) kvm_set_run_in_flight(get(t), 0);
    }
};
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
