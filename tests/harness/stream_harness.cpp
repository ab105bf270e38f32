// CPU harness for the host layer that keeps several calls in flight (kv-match_amd/csrc/kvm_stream.cpp: kvm_run_jobs,
// kvm_run_device_jobs and the persistent worker pool under them).  Compiled by tests/test_abi.py together with
// kvm_stream.cpp itself, once under ThreadSanitizer and once under ASan + UBSan; the library's own entry points the loops
// call (kvm_verify_*, kvm_verify_device, kvm_pack_answers_device, kvm_last_error ...) are stand-ins defined HERE: they
// keep the contract the loops rely on (a context is never entered twice at once; a failing call leaves its text in its
// own context) and sleep instead of launching kernels.  Prints one line per check; exit code = failed checks.
// Argument: "tsan" (threads only) or "fork" (also the fork()ed-child case, which ThreadSanitizer does not support).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>

#include "../../include/kvmatch.h"

struct kvm_ctx {
    std::string err;
    std::atomic<int> inside{0};
    std::atomic<int> calls{0};
    std::atomic<int> in_flight{0}, in_flight_max{0};
};
struct kvm_result {
    int64_t job_tag;
};

static std::atomic<int> g_overlap{0};   // a context entered twice at once
static std::atomic<int64_t> g_started{0};

static int fake_call(kvm_ctx *c, double epsilon, int32_t m, kvm_result **r) {
    if (c->inside.fetch_add(1) != 0) g_overlap++;
    g_started++;
    c->calls++;
    std::this_thread::sleep_for(std::chrono::microseconds(20 + (m % 7) * 15));
    int rc = KVM_OK;
    if (!(epsilon >= 0)) {
        c->err = "epsilon must be >= 0 (job tag " + std::to_string(m) + ")";
        rc = KVM_ERR_ILLEGAL_ARGUMENT;
    } else if (r) {
        *r = new kvm_result{m};
    }
    c->inside.fetch_sub(1);
    return rc;
}

extern "C" {
int kvm_verify_rsm_ed(kvm_ctx *c, const double *, int32_t m, const int32_t *, const int32_t *, int32_t, int32_t, double e, kvm_result **r) { return fake_call(c, e, m, r); }
int kvm_verify_cnsm_ed(kvm_ctx *c, const double *, int32_t m, const int32_t *, const int32_t *, int32_t, int32_t, double e, double, double, kvm_result **r) { return fake_call(c, e, m, r); }
int kvm_verify_rsm_dtw(kvm_ctx *c, const double *, int32_t m, const int32_t *, const int32_t *, int32_t, int32_t, double e, int32_t, kvm_result **r) { return fake_call(c, e, m, r); }
int kvm_verify_cnsm_dtw(kvm_ctx *c, const double *, int32_t m, const int32_t *, const int32_t *, int32_t, int32_t, double e, int32_t, double, double, kvm_result **r) { return fake_call(c, e, m, r); }
int kvm_verify_device(kvm_ctx *c, int32_t, const double *, int32_t m, const int32_t *, const int32_t *, int32_t, int32_t, double e, int32_t, double, double,
                      int32_t *, double *, int64_t, int64_t *, int64_t *cnt, void *) {
    if (cnt) *cnt = m;
    return fake_call(c, e, m, nullptr);
}
int kvm_pack_answers_device(kvm_ctx *, const int32_t *, const double *, const int64_t *, int64_t, double *, void *) { return KVM_OK; }
const char *kvm_last_error(const kvm_ctx *c) { return c ? c->err.c_str() : ""; }
}
static std::atomic<int64_t> g_ready_records{0};
int kvm_record_event_device(kvm_ctx *, void *event, void *) {
    if (event) (*static_cast<std::atomic<int64_t> *>(event))++;
    g_ready_records++;
    return KVM_OK;
}
int kvm_poison_packed_device(kvm_ctx *, double *d_packed, void *) {
    if (d_packed) d_packed[0] = -1.0;
    return KVM_OK;
}
void kvm_set_run_in_flight(kvm_ctx *c, int n) noexcept {
    if (!c) return;
    c->in_flight = n;
    if (n > c->in_flight_max) c->in_flight_max = n;
}
void kvm_set_error_noexcept(kvm_ctx *c, const char *msg) noexcept {
    try {
        if (c) c->err.assign(msg);
    } catch (...) {
    }
}

static int failed = 0;
#define CHECK(cond, ...)                      \
    do {                                      \
        const bool ok_ = (cond);              \
        printf("%s ", ok_ ? "ok  " : "FAIL"); \
        printf(__VA_ARGS__);                  \
        printf("\n");                         \
        if (!ok_) failed++;                   \
    } while (0)

static kvm_job job(int tag, double eps = 1.0) {
    kvm_job j;
    std::memset(&j, 0, sizeof j);
    j.variant = tag % 4;
    j.m = tag;
    j.epsilon = eps;
    return j;
}

// one stream: n_jobs jobs over n_ctx contexts; `bad` (>= 0): that job fails.  Returns the number of violated expectations.
static int one_stream(std::vector<kvm_ctx> &ctxs, int n_jobs, int bad) {
    int wrong = 0;
    std::vector<kvm_ctx *> hs;
    for (auto &c : ctxs) hs.push_back(&c);
    std::vector<kvm_job> jobs;
    for (int i = 0; i < n_jobs; i++) jobs.push_back(job(100 + i, i == bad ? -1.0 : 1.0));
    std::vector<kvm_result *> res((size_t)n_jobs, (kvm_result *)0x1);
    std::vector<int32_t> st((size_t)n_jobs, 77);
    const int rc = kvm_run_jobs(hs.data(), (int32_t)hs.size(), jobs.data(), n_jobs, res.data(), st.data());
    if (bad < 0) {
        if (rc != KVM_OK) wrong++;
        for (int i = 0; i < n_jobs; i++) {
            if (st[(size_t)i] != KVM_OK || !res[(size_t)i] || res[(size_t)i]->job_tag != 100 + i) wrong++;
        }
    } else {
        if (rc != KVM_ERR_ILLEGAL_ARGUMENT || st[(size_t)bad] != KVM_ERR_ILLEGAL_ARGUMENT || res[(size_t)bad] != nullptr) wrong++;
        // the first failing job's text is in the FIRST context, whichever context ran the job
        const std::string want = "epsilon must be >= 0 (job tag " + std::to_string(100 + bad) + ")";
        if (want != kvm_last_error(hs[0])) wrong++;
        // jobs in front of the failing one have all run (they were taken in order); later ones ran or were never started
        for (int i = 0; i < bad; i++)
            if (st[(size_t)i] != KVM_OK || !res[(size_t)i]) wrong++;
        for (int i = bad + 1; i < n_jobs; i++)
            if (st[(size_t)i] != KVM_OK || (res[(size_t)i] && res[(size_t)i]->job_tag != 100 + i)) wrong++;
    }
    for (auto r : res)
        if (r != (kvm_result *)0x1) delete r;
    return wrong;
}

static int device_stream(std::vector<kvm_ctx> &ctxs, int n_jobs, int bad) {
    int wrong = 0;
    const int n_slots = (int)ctxs.size();
    std::vector<kvm_device_slot> slots((size_t)n_slots);
    std::vector<std::vector<double>> packed((size_t)n_slots, std::vector<double>(4, 0.0));
    for (int t = 0; t < n_slots; t++) {
        std::memset(&slots[(size_t)t], 0, sizeof(kvm_device_slot));
        slots[(size_t)t].ctx = &ctxs[(size_t)t];
        slots[(size_t)t].d_packed = packed[(size_t)t].data();
    }
    // (a `ready` event per slot: recorded once per job of the slot, before the job is reported produced)
    std::vector<std::atomic<int64_t>> ready((size_t)n_slots);
    for (int t = 0; t < n_slots; t++) {
        ready[(size_t)t] = 0;
        slots[(size_t)t].ready = &ready[(size_t)t];
    }
    std::vector<kvm_job> jobs;
    for (int i = 0; i < n_jobs; i++) jobs.push_back(job(500 + i, i == bad ? -1.0 : 1.0));
    std::vector<int32_t> produced((size_t)n_jobs, 0), consumed((size_t)n_jobs, 0), st((size_t)n_jobs, 77);
    std::vector<int64_t> cnt((size_t)n_jobs, -1);
    int32_t stop = 0;
    int rc = 12345;
    std::thread producer([&] {
        rc = kvm_run_device_jobs(slots.data(), n_slots, jobs.data(), n_jobs, produced.data(), consumed.data(), &stop, cnt.data(), st.data());
    });
    // the consumer: in query order, as the caller's exchange thread does
    for (int i = 0; i < n_jobs; i++) {
        while (!__atomic_load_n(&produced[(size_t)i], __ATOMIC_ACQUIRE)) std::this_thread::yield();
        const bool after_bad = bad >= 0 && i >= bad && (i - bad) % n_slots == 0;   // the failing slot keeps failing
        if (after_bad ? st[(size_t)i] == KVM_OK : st[(size_t)i] != KVM_OK) wrong++;
        if (!after_bad && cnt[(size_t)i] != 500 + i) wrong++;
        if (ready[(size_t)(i % n_slots)] != i / n_slots + 1) wrong++;   // recorded for job i, and not yet for the slot's next
        __atomic_store_n(&consumed[(size_t)i], 1, __ATOMIC_RELEASE);
    }
    producer.join();
    if (bad < 0 ? rc != KVM_OK : rc != KVM_ERR_ILLEGAL_ARGUMENT) wrong++;
    if (bad >= 0 && packed[(size_t)(bad % n_slots)][0] != -1.0) wrong++;
    return wrong;
}

int main(int argc, char **argv) {
    const bool with_fork = argc > 1 && !std::strcmp(argv[1], "fork");
    // 1. one caller: streams of every shape, with and without a failing job
    {
        std::vector<kvm_ctx> ctxs(8);
        int wrong = 0;
        for (int n_jobs : {0, 1, 3, 8, 9, 50}) wrong += one_stream(ctxs, n_jobs, -1);
        for (int bad : {0, 1, 7, 8, 30, 49}) wrong += one_stream(ctxs, 50, bad);
        CHECK(wrong == 0 && g_overlap == 0, "one caller, 8 contexts: %d violated expectations, %d overlapping entries", wrong, (int)g_overlap);
        // (the contexts are told how many of them the call keeps busy -- 8 for >= 8 jobs, 3 for 3 -- and that it is over)
        int marks = 0;
        for (auto &c : ctxs) marks += c.in_flight == 0 && c.in_flight_max == 8;
        CHECK(marks == 8, "the in-flight mark reaches every context and is taken back (%d of 8)", marks);
        std::vector<kvm_ctx> three(8);
        wrong = one_stream(three, 3, -1);
        CHECK(wrong == 0 && three[0].in_flight_max == 3 && three[7].in_flight_max == 3 && three[7].in_flight == 0, "3 jobs on 8 contexts: 3 in flight");
        std::vector<kvm_ctx> one(1);
        CHECK(one_stream(one, 20, -1) == 0 && one_stream(one, 20, 5) == 0, "one context: the calling thread alone");
        CHECK(one[0].in_flight_max == 1 && one[0].in_flight == 0, "one context: one in flight");
    }
    // 2. several callers at once, each with its own contexts, sharing the pool (what a ShardedStream beside a QueryStream does)
    {
        std::atomic<int> wrong{0};
        std::vector<std::thread> callers;
        for (int c = 0; c < 4; c++)
            callers.emplace_back([&, c] {
                std::vector<kvm_ctx> ctxs((size_t)(2 + 2 * c));
                std::mt19937 rng(17 + c);
                for (int rep = 0; rep < 60; rep++) {
                    const int n_jobs = 1 + (int)(rng() % 40);
                    const int bad = (rng() % 4 == 0) ? (int)(rng() % n_jobs) : -1;
                    wrong += one_stream(ctxs, n_jobs, bad);
                }
            });
        for (auto &t : callers) t.join();
        CHECK(wrong == 0 && g_overlap == 0, "4 callers x 60 streams at once: %d violated expectations, %d overlapping entries", (int)wrong, (int)g_overlap);
    }
    // 3. the device-resident form with its produced / consumed hand-shake
    {
        std::vector<kvm_ctx> ctxs(3);
        int wrong = device_stream(ctxs, 20, -1) + device_stream(ctxs, 20, 7) + device_stream(ctxs, 2, -1);
        CHECK(wrong == 0 && g_overlap == 0, "kvm_run_device_jobs, 3 slots: %d violated expectations", wrong);
        CHECK(ctxs[0].in_flight == 0 && ctxs[2].in_flight_max == 3, "kvm_run_device_jobs marks its slots' contexts (%d)", (int)ctxs[2].in_flight_max);
    }
    // 4. argument checks touch nothing
    {
        kvm_ctx c;
        kvm_ctx *hs[2] = {&c, nullptr};
        kvm_job j = job(1);
        kvm_result *r = nullptr;
        int32_t st = 0;
        CHECK(kvm_run_jobs(hs, 2, &j, 1, &r, &st) == KVM_ERR_INVALID_HANDLE && kvm_run_jobs(nullptr, 1, &j, 1, &r, &st) == KVM_ERR_ILLEGAL_ARGUMENT &&
                  kvm_run_jobs(hs, 1, nullptr, 1, &r, &st) == KVM_ERR_ILLEGAL_ARGUMENT,
              "argument checks");
    }
    // 5. a fork()ed child starts with a pool of its own (the parent's workers do not exist there)
    if (with_fork) {
        fflush(stdout);
        const pid_t pid = fork();
        if (pid == 0) {
            std::vector<kvm_ctx> ctxs(4);
            alarm(20);   // a child that waits for workers it does not have would hang: die instead
            _exit(one_stream(ctxs, 30, -1) == 0 && one_stream(ctxs, 30, 11) == 0 ? 0 : 3);
        }
        int status = -1;
        waitpid(pid, &status, 0);
        CHECK(WIFEXITED(status) && WEXITSTATUS(status) == 0, "fork()ed child runs streams on a pool of its own (status %d)", status);
    }
    printf("%lld calls started\n", (long long)g_started);
    return failed;
}
