// kvm_host_threads.h -- the persistent worker pool under the host layers that keep several calls in flight (kvm_stream.cpp:
// kvm_run_jobs / kvm_run_device_jobs; kvm_sharded.cpp: the slot loops of a device group).  Host only, no HIP.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

#include <pthread.h>

// The helper threads of a call are PERSISTENT (round 5): a process-wide pool of workers that sleep between calls, grown on
// demand and never shrunk.  Rounds 4's form started n_ctx - 1 fresh std::threads per kvm_run_jobs call and joined them at
// its end; every such thread initialises its own HIP thread state on its first launch and tears it down when it exits, and
// a tool that hooks the runtime (rocprofv3's kernel tracing) registers / drops per-thread buffers at the same two points --
// hundreds of times in a bench run.  With workers that live as long as the process, that happens once per worker.
// A call hands its loops to the pool and waits for all of them on every path out (thread_group's destructor): the loops
// reference the call's stack frame.  If a worker cannot be started (std::system_error) the loops already handed over are
// told to stop by the caller's flag and waited for; the C ABI hands back a status.
struct kvm_ctx;
void kvm_set_run_in_flight(kvm_ctx *ctx, int n) noexcept;  // kvm_api.hip (internal)

namespace kvm_host {
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

struct thread_group {
    worker_pool::ticket t;
    void start(std::function<void()> fn) { worker_pool::get().submit(t, std::move(fn)); }
    ~thread_group() { worker_pool::wait(t); }
};

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
}  // namespace kvm_host
