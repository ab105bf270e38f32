// CPU harness for the group layer (kv-match_amd/csrc/kvm_sharded.cpp: kvm_split_intervals, kvm_open_devices / kvm_open_rank,
// kvm_verify_sharded, kvm_run_sharded_jobs).  Compiled by tests/test_sharded.py together with kvm_sharded.cpp itself, once
// under ThreadSanitizer and once under ASan + UBSan.  Everything below the group layer is a stand-in defined HERE:
//   * the device layer (kvm_sharded_dev.h) on host memory -- a stream is a thread executing its queue in order, an event a
//     generation counter, the all-gather a rendezvous of the ranks' stream threads (an in-process group, or several groups
//     opened with kvm_open_rank on the same id and driven by one host thread each: the N > 1 path without a GPU);
//   * the verification path (kvm_verify_device, kvm_pack_answers_device, ...) as a rule that makes a candidate offset an
//     answer from its value alone, so that the merged result of any sharding can be compared with the unsharded list.
// Prints one line per check; exit code = failed checks.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kvmatch.h"
#include "../../kv-match_amd/csrc/kvm_sharded_dev.h"

// ---- streams, events, memory ---------------------------------------------------------------------------------------
namespace {
struct fstream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    uint64_t enq = 0, done = 0;
    bool quit = false;
    std::thread th;
    fstream() {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [&] { return quit || !q.empty(); });
                if (q.empty()) return;
                auto fn = std::move(q.front());
                q.pop_front();
                lk.unlock();
                fn();
                lk.lock();
                done++;
                cv.notify_all();
            }
        });
    }
    void push(std::function<void()> fn) {
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(std::move(fn));
            enq++;
        }
        cv.notify_all();
    }
    void sync() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t want = enq;
        cv.wait(lk, [&] { return done >= want; });
    }
    ~fstream() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_all();
        th.join();
    }
};
struct fevent {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t issued = 0, completed = 0;
};
std::atomic<int64_t> g_live_allocs{0}, g_live_events{0}, g_live_comms{0};
thread_local std::string t_sd_err;
}  // namespace

const char *kvm_sd_last_error() { return t_sd_err.c_str(); }
int kvm_sd_stream_synchronize(int, void *s) {
    ((fstream *)s)->sync();
    return KVM_OK;
}
int kvm_sd_event_create(int, void **e) {
    *e = new fevent();
    g_live_events++;
    return KVM_OK;
}
void kvm_sd_event_destroy(int, void *e) {
    if (e) {
        delete (fevent *)e;
        g_live_events--;
    }
}
int kvm_sd_event_record(int, void *ev, void *s) {
    fevent *e = (fevent *)ev;
    uint64_t gen;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        gen = ++e->issued;
    }
    ((fstream *)s)->push([e, gen] {
        {
            std::lock_guard<std::mutex> lk(e->mu);
            if (e->completed < gen) e->completed = gen;
        }
        e->cv.notify_all();
    });
    return KVM_OK;
}
int kvm_sd_stream_wait_event(int, void *s, void *ev) {
    fevent *e = (fevent *)ev;
    uint64_t gen;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        gen = e->issued;
    }
    ((fstream *)s)->push([e, gen] {
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv.wait(lk, [&] { return e->completed >= gen; });
    });
    return KVM_OK;
}
int kvm_sd_event_synchronize(void *ev) {
    fevent *e = (fevent *)ev;
    std::unique_lock<std::mutex> lk(e->mu);
    const uint64_t gen = e->issued;
    e->cv.wait(lk, [&] { return e->completed >= gen; });
    return KVM_OK;
}
static std::atomic<int64_t> g_fail_malloc_at{-1};  // the n-th device allocation from now fails (-1: never)
int kvm_sd_malloc(int, void **p, size_t bytes) {
    if (g_fail_malloc_at >= 0 && g_fail_malloc_at.fetch_sub(1) == 0) {
        t_sd_err = "hipMalloc: out of memory (injected)";
        *p = nullptr;
        return KVM_ERR_DEVICE;
    }
    *p = std::malloc(bytes ? bytes : 8);
    g_live_allocs++;
    return KVM_OK;
}
void kvm_sd_free(int, void *p) {
    if (p) {
        std::free(p);
        g_live_allocs--;
    }
}
int kvm_sd_host_alloc(int, void **p, size_t bytes) { return kvm_sd_malloc(0, p, bytes); }
void kvm_sd_host_free(void *p) { kvm_sd_free(0, p); }
int kvm_sd_copy_d2h_async(int, void *dst, const void *src, size_t bytes, void *s) {
    ((fstream *)s)->push([dst, src, bytes] { std::memcpy(dst, src, bytes); });
    return KVM_OK;
}
int kvm_sd_readback_blocks(int, double *dst, const double *src, int blocks, size_t bd, void *s) {
    ((fstream *)s)->push([=] {
        for (int b = 0; b < blocks; b++) {
            const double c = src[(size_t)b * bd];
            const size_t cap = bd / 2 - 1, n = c > 0 ? std::min((size_t)c, cap) : 0;
            std::memcpy(dst + (size_t)b * bd, src + (size_t)b * bd, 2 * (n + 1) * sizeof(double));
            // (what lies behind the pairs present is NOT copied: the merge must not read it)
            for (size_t i = 2 * (n + 1); i < bd; i++) dst[(size_t)b * bd + i] = -12345.0;
        }
    });
    return KVM_OK;
}

// ---- the all-gather: a rendezvous of the ranks' stream threads -------------------------------------------------------------
namespace {
struct fworld {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    struct coll {
        std::vector<std::vector<double>> blocks;
        int arrived = 0, left = 0;
    };
    std::map<uint64_t, coll> colls;
};
struct fcomm {
    std::shared_ptr<fworld> w;
    int rank = 0;
    uint64_t seq = 0;
};
std::mutex g_reg_mu;
std::map<std::string, std::weak_ptr<fworld>> g_registry;
std::atomic<uint64_t> g_id_counter{1};
}  // namespace

int kvm_sd_unique_id(void *id, int32_t id_bytes) {
    if (!id || id_bytes != KVM_GROUP_ID_BYTES) return KVM_ERR_ILLEGAL_ARGUMENT;
    std::memset(id, 0, KVM_GROUP_ID_BYTES);
    const uint64_t v = g_id_counter++;
    std::memcpy(id, &v, sizeof v);
    return KVM_OK;
}
int kvm_sd_comm_init_all(const int *, int n, void **comms) {
    auto w = std::make_shared<fworld>();
    w->n = n;
    for (int i = 0; i < n; i++) {
        fcomm *c = new fcomm();
        c->w = w;
        c->rank = i;
        comms[i] = c;
        g_live_comms++;
    }
    return KVM_OK;
}
int kvm_sd_comm_init_rank(int, int rank, int world, const void *id, void **comm) {
    std::shared_ptr<fworld> w;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        const std::string key((const char *)id, KVM_GROUP_ID_BYTES);
        w = g_registry[key].lock();
        if (!w) {
            w = std::make_shared<fworld>();
            w->n = world;
            g_registry[key] = w;
        }
    }
    fcomm *c = new fcomm();
    c->w = w;
    c->rank = rank;
    *comm = c;
    g_live_comms++;
    return KVM_OK;
}
void kvm_sd_comm_destroy(int, void *comm) {
    if (comm) {
        delete (fcomm *)comm;
        g_live_comms--;
    }
}
int kvm_sd_all_gather(int n, const int *, void *const *comms, const void *const *send, void *const *recv, size_t doubles,
                      void *const *streams) {
    for (int i = 0; i < n; i++) {
        fcomm *c = (fcomm *)comms[i];
        const uint64_t seq = c->seq++;
        const double *sp = (const double *)send[i];
        double *rp = (double *)recv[i];
        std::shared_ptr<fworld> w = c->w;
        const int rank = c->rank;
        ((fstream *)streams[i])->push([w, rank, seq, sp, rp, doubles] {
            std::unique_lock<std::mutex> lk(w->mu);
            fworld::coll &k = w->colls[seq];
            if (k.blocks.empty()) k.blocks.resize((size_t)w->n);
            k.blocks[(size_t)rank].assign(sp, sp + doubles);
            k.arrived++;
            w->cv.notify_all();
            w->cv.wait(lk, [&] { return k.arrived == w->n; });
            for (int r = 0; r < w->n; r++) std::memcpy(rp + (size_t)r * doubles, k.blocks[(size_t)r].data(), doubles * 8);
            if (++k.left == w->n) w->colls.erase(seq);
        });
    }
    return KVM_OK;
}

// ---- the verification path -------------------------------------------------------------------------------------------
struct kvm_ctx {
    fstream stream;   // the context's own stream: what the group layer runs the slot on
    int device = 0;
    std::string err;
    std::atomic<int> inside{0};
    bool series = false;
    int in_flight_max = 0;
    long option_sum = 0;
};
struct kvm_result {
    std::vector<int32_t> off;
    std::vector<double> dist;
    int64_t candidates = 0;
};
static std::atomic<int> g_overlap{0};
static std::atomic<int64_t> g_live_ctx{0};
static const int32_t POISON = 777777;  // a shard that holds this offset fails (variant 3 only)

static bool is_answer(int32_t o, int mod) { return mod <= 1 || (((uint32_t)o * 2654435761u) >> 7) % (uint32_t)mod == 0; }
static double dist_of(int32_t o) { return (double)(((uint32_t)o * 40503u) % 1000u) / 8.0; }  // (many ties: the offset decides)

extern "C" {
int kvm_open(int device, kvm_ctx **out) {
    if (device < 0 || device >= 8) return KVM_ERR_ILLEGAL_ARGUMENT;
    *out = new kvm_ctx();
    (*out)->device = device;
    g_live_ctx++;
    return KVM_OK;
}
int kvm_close(kvm_ctx *c) {
    delete c;
    g_live_ctx--;
    return KVM_OK;
}
const char *kvm_last_error(const kvm_ctx *c) { return c ? c->err.c_str() : "open failed (stand-in)"; }
int kvm_load_series_f64(kvm_ctx *c, const double *v, int64_t n) {
    if (!v || n <= 0) {
        c->err = "bad series";
        return KVM_ERR_ILLEGAL_ARGUMENT;
    }
    c->series = true;
    return KVM_OK;
}
int kvm_load_series_be64(kvm_ctx *c, const void *v, int64_t n) { return kvm_load_series_f64(c, (const double *)v, n); }
int kvm_load_series_file(kvm_ctx *c, const char *p, int64_t n) { return kvm_load_series_f64(c, (const double *)p, n); }
int kvm_share_series(kvm_ctx *c, const kvm_ctx *owner) {
    if (!owner->series || owner->device != c->device) return KVM_ERR_NO_SERIES;
    c->series = true;
    return KVM_OK;
}
int kvm_set_option(kvm_ctx *c, int32_t option, int64_t value) {
    c->option_sum += option * 1000 + value;
    return KVM_OK;
}
int kvm_verify_device(kvm_ctx *c, int32_t variant, const double *query, int32_t m, const int32_t *left, const int32_t *right, int32_t k,
                      int32_t, double eps, int32_t, double, double, int32_t *d_off, double *d_dist, int64_t cap, int64_t *d_count,
                      int64_t *cnt, void *stream) {
    if (c->inside.fetch_add(1) != 0) g_overlap++;
    int rc = KVM_OK;
    auto offs = std::make_shared<std::vector<int32_t>>();
    int64_t cc = 0;
    if (!c->series) {
        c->err = "no series loaded";
        rc = KVM_ERR_NO_SERIES;
    } else if (!query || m <= 0 || !(eps >= 0)) {
        c->err = "epsilon must be >= 0 and the query non-empty";
        rc = KVM_ERR_ILLEGAL_ARGUMENT;
    } else {
        const int mod = (int)query[0];
        for (int32_t i = 0; i < k && rc == KVM_OK; i++) {
            cc += (int64_t)right[i] - left[i] + 1;
            for (int64_t o = left[i]; o <= right[i]; o++) {
                if (variant == 3 && o == POISON) {
                    c->err = "left: 777777, length: 1 (the reference's range check; stand-in)";
                    rc = KVM_ERR_IO;
                    break;
                }
                if (is_answer((int32_t)o, mod)) offs->push_back((int32_t)o);
            }
        }
        if (rc == KVM_OK && (int64_t)offs->size() > cap) {
            c->err = "answer buffer too small (the group layer sized it wrongly)";
            rc = KVM_ERR_DEVICE;
        }
    }
    if (rc == KVM_OK) {
        if (cnt) *cnt = cc;
        ((fstream *)stream)->push([offs, d_off, d_dist, d_count] {
            // (unordered on purpose: the device appends answers atomically)
            const size_t n = offs->size();
            for (size_t i = 0; i < n; i++) {
                const size_t j = (i * 7919u) % (n ? n : 1);
                (void)j;
                d_off[i] = (*offs)[n - 1 - i];
                d_dist[i] = dist_of(d_off[i]);
            }
            *d_count = (int64_t)n;
        });
    }
    c->inside.fetch_sub(1);
    return rc;
}
int kvm_pack_answers_device(kvm_ctx *, const int32_t *d_off, const double *d_dist, const int64_t *d_count, int64_t pack_cap, double *d_packed,
                            void *stream) {
    ((fstream *)stream)->push([=] {
        const int64_t n = *d_count;
        d_packed[0] = (double)n;
        d_packed[1] = 0.0;
        for (int64_t i = 0; i < n && i < pack_cap; i++) {
            d_packed[2 * (i + 1)] = (double)d_off[i];
            d_packed[2 * (i + 1) + 1] = d_dist[i];
        }
    });
    return KVM_OK;
}
}
int kvm_poison_packed_device(kvm_ctx *, double *d_packed, void *stream) {
    ((fstream *)stream)->push([d_packed] {
        d_packed[0] = -1.0;
        d_packed[1] = 0.0;
    });
    return KVM_OK;
}
void *kvm_ctx_stream_handle(kvm_ctx *c) { return c ? &c->stream : nullptr; }
void kvm_set_run_in_flight(kvm_ctx *c, int n) noexcept {
    if (c && n > c->in_flight_max) c->in_flight_max = n;
}
kvm_result *kvm_result_from_pairs(const int32_t *off, const double *dist, int64_t cnt, int64_t candidates, double) {
    kvm_result *r = new kvm_result();
    std::vector<size_t> idx((size_t)cnt);
    for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return dist[a] != dist[b] ? dist[a] < dist[b] : off[a] < off[b]; });
    for (size_t i : idx) {
        r->off.push_back(off[i]);
        r->dist.push_back(dist[i]);
    }
    r->candidates = candidates;
    return r;
}

// ---- the checks ---------------------------------------------------------------------------------------------------------
static int failed = 0;
#define CHECK(cond, ...)                      \
    do {                                      \
        const bool ok_ = (cond);              \
        printf("%s ", ok_ ? "ok  " : "FAIL"); \
        printf(__VA_ARGS__);                  \
        printf("\n");                         \
        if (!ok_) failed++;                   \
    } while (0)

struct list_t {
    std::vector<int32_t> left, right;
};
static list_t make_list(std::mt19937 &rng, int k, int kind) {
    list_t l;
    int64_t pos = 1 + (int64_t)(rng() % 50);
    for (int i = 0; i < k; i++) {
        int len = kind == 0 ? 1 + (int)(rng() % 120) : kind == 1 ? (int)(rng() % 5) - 1 : 1 + (int)(rng() % 3000);
        l.left.push_back((int32_t)pos);
        l.right.push_back((int32_t)(pos + len - 1));
        pos += std::max(len, 0) + 1 + (int64_t)(rng() % 40);
    }
    return l;
}
struct expect_t {
    std::vector<int32_t> off;
    std::vector<double> dist;
    int64_t cnt = 0;
};
static expect_t expected(const list_t &l, int mod) {
    expect_t e;
    std::vector<std::pair<double, int32_t>> v;
    for (size_t i = 0; i < l.left.size(); i++) {
        e.cnt += (int64_t)l.right[i] - l.left[i] + 1;
        for (int64_t o = l.left[i]; o <= l.right[i]; o++)
            if (is_answer((int32_t)o, mod)) v.push_back({dist_of((int32_t)o), (int32_t)o});
    }
    std::sort(v.begin(), v.end());
    for (auto &p : v) {
        e.dist.push_back(p.first);
        e.off.push_back(p.second);
    }
    return e;
}
static bool same(const kvm_result *r, const expect_t &e) { return r && r->off == e.off && r->dist == e.dist && r->candidates == e.cnt; }

struct stream_case {
    std::vector<list_t> lists;
    std::vector<std::vector<double>> queries;
    std::vector<kvm_job> jobs;
    std::vector<expect_t> want;
};
static void add_job(stream_case &sc, list_t l, int mod) {
    sc.lists.push_back(std::move(l));
    sc.queries.push_back({(double)mod, 0.0});
    sc.want.push_back(expected(sc.lists.back(), mod));
}
static void finish_case(stream_case &sc, const std::vector<int> &variants, const std::vector<double> &eps) {
    sc.jobs.resize(sc.lists.size());
    for (size_t i = 0; i < sc.lists.size(); i++) {
        kvm_job &j = sc.jobs[i];
        std::memset(&j, 0, sizeof j);
        j.variant = variants.empty() ? (int)(i % 3) : variants[i];
        j.m = 2;
        j.query = sc.queries[i].data();
        j.left = sc.lists[i].left.data();
        j.right = sc.lists[i].right.data();
        j.k = (int32_t)sc.lists[i].left.size();
        j.epsilon = eps.empty() ? 1.0 : eps[i];
    }
}
static stream_case random_case(unsigned seed, int n_jobs, int max_k) {
    std::mt19937 rng(seed);
    stream_case sc;
    for (int i = 0; i < n_jobs; i++) add_job(sc, make_list(rng, (int)(rng() % (unsigned)max_k), (int)(rng() % 3)), 5 + (int)(rng() % 90));
    finish_case(sc, {}, {});
    return sc;
}
static void free_results(std::vector<kvm_result *> &res) {
    for (auto &r : res) {
        delete r;
        r = nullptr;
    }
}
static kvm_group *open_inproc(int n_dev, int F, int flags = 0) {
    std::vector<int32_t> devs;
    for (int i = 0; i < n_dev; i++) devs.push_back(i);
    kvm_group *g = nullptr;
    if (kvm_open_devices(devs.data(), n_dev, F, flags, &g) != KVM_OK) return nullptr;
    static const double series[4] = {1, 2, 3, 4};
    if (kvm_group_load_series_f64(g, series, 4) != KVM_OK) {
        kvm_group_close(g);
        return nullptr;
    }
    return g;
}
static int run_and_compare(kvm_group *g, stream_case &sc) {
    const int64_t n = (int64_t)sc.jobs.size();
    std::vector<kvm_result *> res((size_t)n, (kvm_result *)0x1);
    std::vector<int32_t> st((size_t)n, 77);
    int wrong = 0;
    if (kvm_run_sharded_jobs(g, sc.jobs.data(), n, res.data(), st.data()) != KVM_OK) wrong++;
    for (int64_t i = 0; i < n; i++)
        if (st[(size_t)i] != KVM_OK || res[(size_t)i] == (kvm_result *)0x1 || !same(res[(size_t)i], sc.want[(size_t)i])) wrong++;
    for (auto &r : res)
        if (r == (kvm_result *)0x1) r = nullptr;
    free_results(res);
    return wrong;
}

int main() {
    // 1. the split rule: every candidate of the list in exactly one shard, whatever the mode; interleaved keeps whole intervals
    {
        std::mt19937 rng(11);
        int wrong = 0;
        for (int trial = 0; trial < 400; trial++) {
            const list_t l = make_list(rng, (int)(rng() % 300), (int)(rng() % 3));
            const int parts = 1 + (int)(rng() % 9), mode = (int)(rng() % 3);
            std::map<int32_t, int> seen;
            int64_t total = 0, mx = 0, mn = INT64_MAX;
            for (int p = 0; p < parts; p++) {
                std::vector<int32_t> ol(l.left.size() + 1), orr(l.left.size() + 1);
                const int64_t k = kvm_split_intervals(l.left.data(), l.right.data(), (int64_t)l.left.size(), parts, p, mode, ol.data(), orr.data());
                if (k < 0 || k != kvm_split_intervals(l.left.data(), l.right.data(), (int64_t)l.left.size(), parts, p, mode, nullptr, nullptr)) wrong++;
                int64_t mine = 0;
                for (int64_t i = 0; i < k; i++) {
                    if (orr[(size_t)i] < ol[(size_t)i]) wrong++;
                    for (int64_t o = ol[(size_t)i]; o <= orr[(size_t)i]; o++) seen[(int32_t)o]++;
                    mine += (int64_t)orr[(size_t)i] - ol[(size_t)i] + 1;
                }
                total += mine;
                mx = std::max(mx, mine);
                mn = std::min(mn, mine);
            }
            int64_t want = 0, longest = 0;
            for (size_t i = 0; i < l.left.size(); i++) {
                const int64_t c = std::max<int64_t>(0, (int64_t)l.right[i] - l.left[i] + 1);
                want += c;
                longest = std::max(longest, c);
                for (int64_t o = l.left[i]; o <= l.right[i]; o++)
                    if (seen[(int32_t)o] != 1) wrong++;
            }
            if (total != want || (int64_t)seen.size() != want) wrong++;
            // balance: contiguous within one candidate, interleaved within one interval length per round ... loosely: 2 longest
            if (mode == KVM_SPLIT_CONTIGUOUS && mx - mn > 1) wrong++;
        }
        CHECK(wrong == 0, "kvm_split_intervals: 400 lists x parts 1..9 x 3 modes partition their candidates (%d violations)", wrong);
        CHECK(kvm_split_intervals(nullptr, nullptr, 3, 2, 0, 0, nullptr, nullptr) == KVM_ERR_ILLEGAL_ARGUMENT &&
                  kvm_split_intervals(nullptr, nullptr, 0, 2, 2, 0, nullptr, nullptr) == KVM_ERR_ILLEGAL_ARGUMENT &&
                  kvm_split_intervals(nullptr, nullptr, 0, 2, 1, 0, nullptr, nullptr) == 0,
              "kvm_split_intervals: argument checks");
    }
    // 2. one process, n_dev devices: streams of random jobs == the unsharded answers
    for (int n_dev : {1, 2, 4, 8})
        for (int F : {1, 3}) {
            kvm_group *g = open_inproc(n_dev, F, n_dev == 1 && F == 3 ? KVM_GROUP_RCCL_ALWAYS : 0);
            int wrong = g ? 0 : 1;
            if (g) {
                for (unsigned seed : {1u, 2u}) {
                    stream_case sc = random_case(seed * 100 + (unsigned)n_dev, seed == 1 ? 25 : 4, 160);
                    wrong += run_and_compare(g, sc);
                }
                stream_case none = random_case(5, 0, 10);
                wrong += run_and_compare(g, none);
                if (kvm_group_world(g) != n_dev || kvm_group_local_devices(g) != n_dev ||
                    kvm_group_rccl_ranks(g) != ((n_dev > 1 || F == 3) ? n_dev : 0))
                    wrong++;
                kvm_group_close(g);
            }
            CHECK(wrong == 0 && g_overlap == 0, "in-process group, %d device(s), %d in flight: %d violated expectations", n_dev, F, wrong);
        }
    // 3. one process per rank (a thread each here): every rank ends with the full list
    for (int world : {2, 3}) {
        uint8_t id[KVM_GROUP_ID_BYTES];
        kvm_group_unique_id(id, KVM_GROUP_ID_BYTES);
        std::atomic<int> wrong{0};
        std::vector<std::thread> ranks;
        for (int r = 0; r < world; r++)
            ranks.emplace_back([&, r] {
                kvm_group *g = nullptr;
                static const double series[4] = {1, 2, 3, 4};
                if (kvm_open_rank(r % 8, r, world, id, KVM_GROUP_ID_BYTES, 2, 0, &g) != KVM_OK || kvm_group_load_series_f64(g, series, 4) != KVM_OK) {
                    wrong++;
                    return;
                }
                stream_case sc = random_case(900 + (unsigned)world, 30, 120);   // (the same jobs on every rank)
                wrong += run_and_compare(g, sc);
                if (kvm_group_world(g) != world || kvm_group_local_devices(g) != 1 || kvm_group_rccl_ranks(g) != world) wrong++;
                kvm_group_close(g);
            });
        for (auto &t : ranks) t.join();
        CHECK(wrong == 0, "%d ranks, one group each on one id: every rank holds the merged answers (%d violations)", world, (int)wrong);
    }
    // 4. more answers than the fixed-size block holds: the exact-size second exchange
    {
        kvm_group *g = open_inproc(4, 2);
        int wrong = g ? 0 : 1;
        if (g) {
            if (kvm_group_configure(g, KVM_GROUP_PACK_CAP, 8) != KVM_OK) wrong++;
            std::mt19937 rng(3);
            stream_case sc;
            for (int i = 0; i < 12; i++) add_job(sc, make_list(rng, 40, 0), i % 3 == 1 ? 1 : 7);  // (mod 1: every candidate answers)
            finish_case(sc, {}, {});
            wrong += run_and_compare(g, sc);
            kvm_group_close(g);
        }
        uint8_t id[KVM_GROUP_ID_BYTES];
        kvm_group_unique_id(id, KVM_GROUP_ID_BYTES);
        std::atomic<int> rw{0};
        std::vector<std::thread> ranks;
        for (int r = 0; r < 2; r++)
            ranks.emplace_back([&, r] {
                kvm_group *g2 = nullptr;
                static const double series[4] = {1, 2, 3, 4};
                if (kvm_open_rank(r, r, 2, id, KVM_GROUP_ID_BYTES, 3, 0, &g2) || kvm_group_load_series_f64(g2, series, 4) ||
                    kvm_group_configure(g2, KVM_GROUP_PACK_CAP, 5)) {
                    rw++;
                    return;
                }
                std::mt19937 rng(4);
                stream_case sc;
                for (int i = 0; i < 10; i++) add_job(sc, make_list(rng, 30, 0), i % 2 ? 1 : 11);
                finish_case(sc, {}, {});
                rw += run_and_compare(g2, sc);
                kvm_group_close(g2);
            });
        for (auto &t : ranks) t.join();
        CHECK(wrong == 0 && rw == 0, "overflow of the packed block: exact-size exchange, in-process (%d) and across ranks (%d)", wrong, (int)rw);
    }
    // 5. a failing shard: in one process the call returns that shard's own status and text; across ranks the owner returns its own,
    //    the others KVM_ERR_SHARD_FAILED, all at the same job; the jobs in front are complete, none behind is returned; the groups
    //    stay usable
    {
        auto failing_case = [](int bad_at, int n_jobs) {
            std::mt19937 rng(77);
            stream_case sc;
            std::vector<int> variants;
            for (int i = 0; i < n_jobs; i++) {
                list_t l = make_list(rng, 60, 0);
                if (i == bad_at) {   // one interval far away that holds the poisoned offset: exactly one shard fails
                    l.left.push_back(POISON - 3);
                    l.right.push_back(POISON + 2);
                }
                add_job(sc, std::move(l), 13);
                variants.push_back(i == bad_at ? 3 : i % 3);
            }
            finish_case(sc, variants, {});
            return sc;
        };
        int wrong = 0;
        kvm_group *g = open_inproc(4, 3);
        for (int bad_at : {0, 1, 5, 11}) {
            stream_case sc = failing_case(bad_at, 12);
            std::vector<kvm_result *> res(12, nullptr);
            std::vector<int32_t> st(12, 77);
            const int rc = kvm_run_sharded_jobs(g, sc.jobs.data(), 12, res.data(), st.data());
            if (rc != KVM_ERR_IO || st[(size_t)bad_at] != KVM_ERR_IO || std::string(kvm_group_last_error(g)).find("777777") == std::string::npos) wrong++;
            for (int i = 0; i < 12; i++) {
                if (i < bad_at && (st[(size_t)i] != KVM_OK || !same(res[(size_t)i], sc.want[(size_t)i]))) wrong++;
                if (i >= bad_at && res[(size_t)i]) wrong++;
            }
            free_results(res);
            stream_case ok = random_case(31, 7, 90);
            wrong += run_and_compare(g, ok);
        }
        kvm_group_close(g);
        CHECK(wrong == 0, "a failing shard inside one process: its status and text, the stream ends there, the group stays usable (%d)", wrong);

        uint8_t id[KVM_GROUP_ID_BYTES];
        kvm_group_unique_id(id, KVM_GROUP_ID_BYTES);
        std::atomic<int> rw{0}, owners{0}, others{0};
        std::vector<std::thread> ranks;
        for (int r = 0; r < 3; r++)
            ranks.emplace_back([&, r] {
                kvm_group *g2 = nullptr;
                static const double series[4] = {1, 2, 3, 4};
                if (kvm_open_rank(r, r, 3, id, KVM_GROUP_ID_BYTES, 2, 0, &g2) || kvm_group_load_series_f64(g2, series, 4)) {
                    rw++;
                    return;
                }
                for (int bad_at : {0, 4, 8}) {
                    stream_case sc = failing_case(bad_at, 9);
                    std::vector<kvm_result *> res(9, nullptr);
                    std::vector<int32_t> st(9, 77);
                    const int rc = kvm_run_sharded_jobs(g2, sc.jobs.data(), 9, res.data(), st.data());
                    if (rc == KVM_ERR_IO) owners++;
                    else if (rc == KVM_ERR_SHARD_FAILED && std::string(kvm_group_last_error(g2)).find("rank(s)") != std::string::npos) others++;
                    else rw++;
                    if (st[(size_t)bad_at] != rc) rw++;
                    for (int i = 0; i < 9; i++) {
                        if (i < bad_at && !same(res[(size_t)i], sc.want[(size_t)i])) rw++;
                        if (i >= bad_at && res[(size_t)i]) rw++;
                    }
                    free_results(res);
                    stream_case ok = random_case(41, 5, 70);
                    rw += run_and_compare(g2, ok);
                }
                kvm_group_close(g2);
            });
        for (auto &t : ranks) t.join();
        CHECK(rw == 0 && owners == 3 && others == 6, "a failing shard across 3 ranks: the owner's status, KVM_ERR_SHARD_FAILED on the others (%d, %d, %d)",
              (int)rw, (int)owners, (int)others);
    }
    // 6. a job every shard refuses (bad epsilon), a malformed job (null arrays), no series, bad arguments
    {
        int wrong = 0;
        kvm_group *g = open_inproc(2, 2);
        std::mt19937 rng(5);
        stream_case sc;
        for (int i = 0; i < 4; i++) add_job(sc, make_list(rng, 50, 0), 9);
        finish_case(sc, {}, {1.0, 1.0, -1.0, 1.0});
        std::vector<kvm_result *> res(4, nullptr);
        std::vector<int32_t> st(4, 77);
        if (kvm_run_sharded_jobs(g, sc.jobs.data(), 4, res.data(), st.data()) != KVM_ERR_ILLEGAL_ARGUMENT || st[2] != KVM_ERR_ILLEGAL_ARGUMENT || res[2] ||
            res[3] || !same(res[0], sc.want[0]) || !same(res[1], sc.want[1]))
            wrong++;
        free_results(res);
        sc.jobs[2].epsilon = 1.0;
        sc.jobs[1].left = nullptr;
        if (kvm_run_sharded_jobs(g, sc.jobs.data(), 4, res.data(), st.data()) != KVM_ERR_ILLEGAL_ARGUMENT || st[1] != KVM_ERR_ILLEGAL_ARGUMENT || !res[0] || res[1])
            wrong++;
        free_results(res);
        kvm_result *one = nullptr;
        sc.jobs[1].left = sc.lists[1].left.data();
        const kvm_job &j = sc.jobs[3];
        if (kvm_verify_sharded(g, j.variant, j.query, j.m, j.left, j.right, j.k, 0, 1.0, 0, 0, 0, &one) != KVM_OK || !same(one, sc.want[3])) wrong++;
        delete one;
        if (kvm_verify_sharded(g, 0, j.query, j.m, j.left, j.right, j.k, 0, 1.0, 0, 0, 0, nullptr) != KVM_ERR_ILLEGAL_ARGUMENT) wrong++;
        if (kvm_run_sharded_jobs(nullptr, sc.jobs.data(), 4, res.data(), st.data()) != KVM_ERR_INVALID_HANDLE) wrong++;
        if (kvm_group_set_option(g, 4, 1) != KVM_OK || kvm_group_ctx(g, 1, 1) == nullptr || kvm_group_ctx(g, 2, 0) != nullptr || kvm_group_ctx(g, 0, 2) != nullptr) wrong++;
        if (kvm_group_configure(g, KVM_GROUP_SPLIT_MODE, KVM_SPLIT_CONTIGUOUS) != KVM_OK || kvm_group_configure(g, KVM_GROUP_SPLIT_MODE, 9) != KVM_ERR_ILLEGAL_ARGUMENT ||
            kvm_group_configure(g, 99, 1) != KVM_ERR_ILLEGAL_ARGUMENT || kvm_group_configure(g, KVM_GROUP_PACK_CAP, 0) != KVM_ERR_ILLEGAL_ARGUMENT)
            wrong++;
        stream_case contiguous = random_case(61, 9, 100);   // (contiguous slices cut inside intervals: still the same answers)
        wrong += run_and_compare(g, contiguous);
        kvm_group_close(g);
        std::vector<int32_t> devs = {0, 1};
        kvm_group *ns = nullptr;
        if (kvm_open_devices(devs.data(), 2, 1, 0, &ns) != KVM_OK) wrong++;
        if (ns && kvm_run_sharded_jobs(ns, sc.jobs.data(), 4, res.data(), st.data()) != KVM_ERR_NO_SERIES) wrong++;
        if (ns) kvm_group_close(ns);
        std::vector<int32_t> twice = {1, 1};
        kvm_group *bad = (kvm_group *)0x1;
        if (kvm_open_devices(twice.data(), 2, 1, 0, &bad) != KVM_ERR_ILLEGAL_ARGUMENT || bad != nullptr) wrong++;
        if (kvm_open_devices(devs.data(), 2, 0, 0, &bad) != KVM_ERR_ILLEGAL_ARGUMENT || kvm_open_devices(devs.data(), 2, 1, 8, &bad) != KVM_ERR_ILLEGAL_ARGUMENT) wrong++;
        if (kvm_open_rank(0, 2, 2, devs.data(), KVM_GROUP_ID_BYTES, 1, 0, &bad) != KVM_ERR_ILLEGAL_ARGUMENT || kvm_open_rank(0, 0, 2, nullptr, 0, 1, 0, &bad) != KVM_ERR_ILLEGAL_ARGUMENT)
            wrong++;
        std::vector<int32_t> nine = {0, 9};   // (device 9 does not exist: the open fails half-way and gives everything back)
        if (kvm_open_devices(nine.data(), 2, 2, 0, &bad) == KVM_OK || std::string(kvm_group_last_error(nullptr)).empty()) wrong++;
        CHECK(wrong == 0, "refused jobs, malformed jobs, no series, bad arguments, a half-opened group (%d)", wrong);
    }
    // 7. device allocations that fail: at open, and while a slot grows its answer arrays in the middle of a stream
    {
        int wrong = 0;
        for (int at : {0, 3, 9, 13}) {   // (a group of 2 devices x 2 slots makes 14 allocations)
            g_fail_malloc_at = at;
            std::vector<int32_t> devs = {0, 1};
            kvm_group *g = nullptr;
            if (kvm_open_devices(devs.data(), 2, 2, 0, &g) != KVM_ERR_DEVICE || g) wrong++;
            g_fail_malloc_at = -1;
        }
        kvm_group *g = open_inproc(2, 2);
        stream_case sc = random_case(71, 6, 100);
        g_fail_malloc_at = 1;   // (the second allocation of the stream: some slot's answer arrays)
        std::vector<kvm_result *> res(6, nullptr);
        std::vector<int32_t> st(6, 77);
        const int rc = kvm_run_sharded_jobs(g, sc.jobs.data(), 6, res.data(), st.data());
        g_fail_malloc_at = -1;
        if (rc != KVM_ERR_DEVICE || std::string(kvm_group_last_error(g)).find("injected") == std::string::npos) wrong++;
        free_results(res);
        wrong += run_and_compare(g, sc);
        kvm_group_close(g);
        CHECK(wrong == 0, "failing device allocations at open and in a stream: a status, nothing leaked, the group stays usable (%d)", wrong);
    }
    CHECK(g_live_allocs == 0 && g_live_events == 0 && g_live_comms == 0 && g_live_ctx == 0 && g_overlap == 0,
          "everything given back: %lld allocations, %lld events, %lld communicators, %lld contexts; %d overlapping entries",
          (long long)g_live_allocs, (long long)g_live_events, (long long)g_live_comms, (long long)g_live_ctx, (int)g_overlap);
    return failed;
}
