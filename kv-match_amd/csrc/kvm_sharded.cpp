// kvm_sharded.cpp -- one engine over the GPUs of a node (include/kvmatch.h: kvm_group): the split rule, the slot loops, the
// exchange protocol around ONE all-gather per query, merge and failure handling.  Host only: everything it needs from HIP and
// RCCL is behind kvm_sharded_dev.h, everything it needs from the verification path is the public C ABI (kvm_verify_device,
// kvm_pack_answers_device, ...), so tests/harness/sharded_harness.cpp runs this file as it is, over stand-ins, under
// ThreadSanitizer and ASan.
//
// The reference's phase 2 walks the interval list once per query (K/QueryEngine.java:343-363 and the same loop in the other
// three engines); candidates are independent given the series and the query, so the list shards freely.  Threads
// of a call: one per (local device, slot) taken from the persistent pool of kvm_host_threads.h -- they split, enqueue, never
// wait for a kernel (except the one read a DTW call makes), and whichever of them completes a job's last local part issues
// that job's collective, in job order on every rank (a turnstile) -- and the CALLING thread, which waits for each query's
// block and merges it (protocol: above run_state, below).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/kvmatch.h"
#include "kvm_host_threads.h"
#include "kvm_sharded_dev.h"

// internal entry points of the library (kvm_verify.hip)
int kvm_poison_packed_device(kvm_ctx *ctx, double *d_packed, void *stream);
void *kvm_ctx_stream_handle(kvm_ctx *ctx);  // kvm_api.hip: the context's own stream
// answers.sort(Comparator.comparing(Pair::getSecond)) on an offset-ascending list (K/QueryEngine.java:373) == (distance, offset)
kvm_result *kvm_result_from_pairs(const int32_t *off, const double *dist, int64_t cnt, int64_t candidates, double elapsed_ms);

// ---- the split rule ---------------------------------------------------------------------------------------------
namespace {
inline int64_t cands_of(int32_t l, int32_t r) {
    const int64_t c = (int64_t)r - (int64_t)l + 1;
    return c > 0 ? c : 0;
}
inline int resolve_mode(int mode, int64_t k, int parts) {
    if (mode != KVM_SPLIT_AUTO) return mode;
    return k >= 8 * (int64_t)parts ? KVM_SPLIT_INTERLEAVED : KVM_SPLIT_CONTIGUOUS;
}
constexpr int MAX_PARTS = 256;

// Interleaved deal: owner[i] = the part interval i goes to.  Rounds of `parts` consecutive intervals; inside a round the
// intervals sorted by length (descending, stable) meet the parts sorted by load so far (ascending, stable): the longest goes
// to the lightest.  Keys are packed so that a plain sort IS the stable one: (count << 8 | 255 - j) descending for the
// intervals, (load << 8 | part) ascending for the parts.
void deal_owners(const int32_t *left, const int32_t *right, int64_t k, int parts, uint8_t *owner) {
    uint64_t load[MAX_PARTS] = {0}, ikey[MAX_PARTS], pkey[MAX_PARTS];
    for (int64_t base = 0; base < k; base += parts) {
        const int len = (int)std::min<int64_t>(parts, k - base);
        for (int j = 0; j < len; j++) {   // insertion sort, descending (rounds are short: parts <= 8 on one node)
            const uint64_t key = ((uint64_t)cands_of(left[base + j], right[base + j]) << 8) | (uint64_t)(255 - j);
            int p = j;
            for (; p > 0 && ikey[p - 1] < key; p--) ikey[p] = ikey[p - 1];
            ikey[p] = key;
        }
        for (int r = 0; r < parts; r++) {  // ascending
            const uint64_t key = (load[r] << 8) | (uint64_t)r;
            int p = r;
            for (; p > 0 && pkey[p - 1] > key; p--) pkey[p] = pkey[p - 1];
            pkey[p] = key;
        }
        for (int j = 0; j < len; j++) {
            const int idx = 255 - (int)(ikey[j] & 255), part = (int)(pkey[j] & 255);
            owner[base + idx] = (uint8_t)part;
            load[part] += ikey[j] >> 8;
        }
    }
}

// shards first .. first + count - 1 of `parts`, appended to L[p - first] / R[p - first]
void split_parts(const int32_t *left, const int32_t *right, int64_t k, int parts, int mode, int first, int count,
                 std::vector<std::vector<int32_t>> &L, std::vector<std::vector<int32_t>> &R) {
    L.assign((size_t)count, {});
    R.assign((size_t)count, {});
    mode = resolve_mode(mode, k, parts);
    if (mode == KVM_SPLIT_INTERLEAVED) {
        if (parts == 1) {
            L[0].reserve((size_t)k);
            R[0].reserve((size_t)k);
            for (int64_t i = 0; i < k; i++)
                if (right[i] >= left[i]) {
                    L[0].push_back(left[i]);
                    R[0].push_back(right[i]);
                }
            return;
        }
        std::vector<uint8_t> owner((size_t)k);
        deal_owners(left, right, k, parts, owner.data());
        for (int c = 0; c < count; c++) {
            L[(size_t)c].reserve((size_t)(k / parts + parts));
            R[(size_t)c].reserve((size_t)(k / parts + parts));
        }
        for (int64_t i = 0; i < k; i++) {
            const int c = (int)owner[(size_t)i] - first;
            if (c >= 0 && c < count && right[i] >= left[i]) {
                L[(size_t)c].push_back(left[i]);
                R[(size_t)c].push_back(right[i]);
            }
        }
        return;
    }
    // contiguous slices of the candidate sequence
    std::vector<int64_t> cum((size_t)k + 1);
    cum[0] = 0;
    for (int64_t i = 0; i < k; i++) cum[(size_t)i + 1] = cum[(size_t)i] + cands_of(left[i], right[i]);
    const __int128 total = cum[(size_t)k];
    for (int c = 0; c < count; c++) {
        const int part = first + c;
        const int64_t lo = (int64_t)(total * part / parts), hi = (int64_t)(total * (part + 1) / parts);
        if (hi <= lo) continue;
        // a = the last index with cum[a] <= lo, b = the last index with cum[b] < hi
        const int64_t a = (std::upper_bound(cum.begin(), cum.end(), lo) - cum.begin()) - 1;
        const int64_t b = (std::lower_bound(cum.begin(), cum.end(), hi) - cum.begin()) - 1;
        for (int64_t i = a; i <= b; i++) {
            int64_t l = left[i], r = right[i];
            if (i == a) l = (int64_t)left[i] + (lo - cum[(size_t)a]);
            if (i == b) r = (int64_t)left[i] + (hi - cum[(size_t)b]) - 1;
            if (r >= l) {
                L[(size_t)c].push_back((int32_t)l);
                R[(size_t)c].push_back((int32_t)r);
            }
        }
    }
}

thread_local std::string t_open_err;
int open_fail(int code, const std::string &msg) {
    try {
        t_open_err = msg;
    } catch (...) {
    }
    return code;
}
}  // namespace

extern "C" int64_t kvm_split_intervals(const int32_t *left, const int32_t *right, int64_t k, int32_t parts, int32_t part,
                                       int32_t mode, int32_t *out_left, int32_t *out_right) {
    if (k < 0 || (k > 0 && (!left || !right)) || parts < 1 || parts > MAX_PARTS || part < 0 || part >= parts || mode < 0 ||
        mode > KVM_SPLIT_CONTIGUOUS || ((out_left == nullptr) != (out_right == nullptr)))
        return KVM_ERR_ILLEGAL_ARGUMENT;
    try {
        std::vector<std::vector<int32_t>> L, R;
        split_parts(left, right, k, parts, mode, part, 1, L, R);
        if (out_left && !L[0].empty()) {
            std::memcpy(out_left, L[0].data(), L[0].size() * sizeof(int32_t));
            std::memcpy(out_right, R[0].data(), R[0].size() * sizeof(int32_t));
        }
        return (int64_t)L[0].size();
    } catch (...) {
        return KVM_ERR_NO_MEMORY;
    }
}

extern "C" int32_t kvm_abi_version(void) { return KVM_ABI_VERSION; }

// ---- the group ----------------------------------------------------------------------------------------------------
namespace {
struct gslot_t {
    kvm_ctx *ctx = nullptr;
    void *stream = nullptr;  // the context's own (not owned here): the slot's kernels AND its job's exchange run on it
    int32_t *d_off = nullptr;
    double *d_dist = nullptr;
    int64_t cap = 0;
    int64_t *d_count = nullptr;
    double *d_packed = nullptr;    // 2 * (pack_cap + 1) doubles: {count, 0}, then (offset, distance) pairs
    double *d_gathered = nullptr;  // world blocks of that size
};
struct gdev_t {
    int device = -1;
    void *comm = nullptr;
    void *gathered = nullptr;   // event behind this device's most recent all-gather (the next one, on another stream, waits for it)
    std::vector<gslot_t> slots;
};
}  // namespace

struct kvm_group {
    int world = 1, rank0 = 0;  // ranks of the exchange; the rank of local device 0 (local device d is rank rank0 + d)
    int in_flight = 1;
    int64_t pack_cap = 4096;
    int split_mode = KVM_SPLIT_AUTO;
    bool use_rccl = false;
    bool loaded = false;
    std::vector<gdev_t> devs;
    std::vector<double *> h_block;  // per slot index: pinned, world blocks (what the exchange of that slot's job lands in)
    std::vector<void *> issued;     // per slot index: event on that slot's stream of devs[0], behind the readback
    std::string err;
    std::atomic<int64_t> stat[KVM_GROUP_STAT_COUNT] = {};  // where the threads of the last stream spent their time (kvm_group_stat)
};

namespace {
int gfail(kvm_group *g, int code, const std::string &msg) {
    try {
        g->err = msg;
    } catch (...) {
    }
    return code;
}
int gfail_sd(kvm_group *g, int code) { return gfail(g, code, kvm_sd_last_error()); }
#define GCHK(g, call)                        \
    do {                                     \
        int rc_ = (call);                    \
        if (rc_ != KVM_OK) return gfail_sd(g, rc_); \
    } while (0)

size_t block_doubles(const kvm_group *g) { return 2 * ((size_t)g->pack_cap + 1); }

void free_exchange(kvm_group *g) {
    for (gdev_t &d : g->devs)
        for (gslot_t &s : d.slots) {
            kvm_sd_free(d.device, s.d_packed);
            kvm_sd_free(d.device, s.d_gathered);
            s.d_packed = s.d_gathered = nullptr;
        }
    for (double *&p : g->h_block) {
        kvm_sd_host_free(p);
        p = nullptr;
    }
}
int alloc_exchange(kvm_group *g) {
    const size_t bd = block_doubles(g);
    for (gdev_t &d : g->devs)
        for (gslot_t &s : d.slots) {
            GCHK(g, kvm_sd_malloc(d.device, (void **)&s.d_packed, bd * 8));
            if (g->use_rccl) GCHK(g, kvm_sd_malloc(d.device, (void **)&s.d_gathered, bd * 8 * (size_t)g->world));
        }
    for (double *&p : g->h_block) GCHK(g, kvm_sd_host_alloc(g->devs[0].device, (void **)&p, bd * 8 * (size_t)g->world));
    return KVM_OK;
}
void destroy_group(kvm_group *g) {
    if (!g) return;
    for (gdev_t &d : g->devs)  // nothing of a finished call is still running (every call drains its streams), but a failed one may be
        for (gslot_t &s : d.slots)
            if (s.stream) (void)kvm_sd_stream_synchronize(d.device, s.stream);
    free_exchange(g);
    for (size_t t = 0; t < g->issued.size(); t++)
        if (!g->devs.empty()) kvm_sd_event_destroy(g->devs[0].device, g->issued[t]);
    for (gdev_t &d : g->devs) {
        if (d.comm) kvm_sd_comm_destroy(d.device, d.comm);
        kvm_sd_event_destroy(d.device, d.gathered);
        for (size_t t = d.slots.size(); t-- > 0;) {  // (slot 0 owns the series the others share: closed last)
            gslot_t &s = d.slots[t];
            kvm_sd_free(d.device, s.d_off);
            kvm_sd_free(d.device, s.d_dist);
            kvm_sd_free(d.device, s.d_count);
            if (s.ctx) kvm_close(s.ctx);   // (with its stream)
        }
    }
    delete g;
}

int open_common(const int32_t *devices, int32_t n_dev, int32_t rank0, int32_t world, const void *id, bool in_process,
                int32_t in_flight, int32_t flags, kvm_group **out) {
    if (!out) return open_fail(KVM_ERR_ILLEGAL_ARGUMENT, "null group pointer");
    *out = nullptr;
    if (!devices || n_dev < 1 || n_dev > MAX_PARTS || world < n_dev || world > MAX_PARTS || in_flight < 1 || in_flight > 64 ||
        rank0 < 0 || rank0 + n_dev > world || (flags & ~KVM_GROUP_RCCL_ALWAYS))
        return open_fail(KVM_ERR_ILLEGAL_ARGUMENT, "bad device list, rank, world, in_flight (1 .. 64) or flags");
    for (int i = 0; i < n_dev; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) return open_fail(KVM_ERR_ILLEGAL_ARGUMENT, "a device is listed twice (RCCL wants one rank per GPU)");
    kvm_group *g = new kvm_group();
    auto bail = [&](int code, std::string msg) {   // (by value: the text may live in the group that is destroyed here)
        destroy_group(g);
        return open_fail(code, msg);
    };
    g->world = world;
    g->rank0 = rank0;
    g->in_flight = in_flight;
    g->use_rccl = world > 1 || (flags & KVM_GROUP_RCCL_ALWAYS);
    g->devs.resize((size_t)n_dev);
    g->h_block.assign((size_t)in_flight, nullptr);
    g->issued.assign((size_t)in_flight, nullptr);
    for (int d = 0; d < n_dev; d++) {
        gdev_t &dv = g->devs[(size_t)d];
        dv.device = devices[d];
        dv.slots.resize((size_t)in_flight);
        for (gslot_t &s : dv.slots) {
            int rc = kvm_open(dv.device, &s.ctx);
            if (rc) return bail(rc, kvm_last_error(nullptr));
            // A slot runs on its context's OWN stream, and so does its job's exchange.  The runtime deals streams out to its few
            // hardware queues (four) in the order they are created, and the streams of one queue run one behind the other: a
            // second stream per slot, created right behind the context's, put every slot stream on the same two queues (an
            // early-abandoning stream ran at 72 us per query instead of 51), and a separate exchange stream queues a query's
            // two tiny operations behind whatever 80-us pruner launch is at the head of its queue (36 us per query; at the
            // highest priority it is worse, not better).  profiles/r06_group_stream_probe.log
            s.stream = kvm_ctx_stream_handle(s.ctx);
            if ((rc = kvm_sd_malloc(dv.device, (void **)&s.d_count, 8))) return bail(rc, kvm_sd_last_error());
        }
    }
    for (void *&e : g->issued)
        if (int rc = kvm_sd_event_create(g->devs[0].device, &e)) return bail(rc, kvm_sd_last_error());
    for (gdev_t &dv : g->devs)
        if (int rc = kvm_sd_event_create(dv.device, &dv.gathered)) return bail(rc, kvm_sd_last_error());
    if (int rc = alloc_exchange(g)) return bail(rc, g->err);
    if (g->use_rccl) {
        int rc;
        if (in_process) {
            std::vector<int> devs(devices, devices + n_dev);
            std::vector<void *> comms((size_t)n_dev, nullptr);
            rc = kvm_sd_comm_init_all(devs.data(), n_dev, comms.data());
            for (int d = 0; d < n_dev && rc == KVM_OK; d++) g->devs[(size_t)d].comm = comms[(size_t)d];
        } else {
            rc = kvm_sd_comm_init_rank(devices[0], rank0, world, id, &g->devs[0].comm);
        }
        if (rc) return bail(rc, kvm_sd_last_error());
    }
    *out = g;
    return KVM_OK;
}

template <typename F>
int guarded_open(F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return open_fail(KVM_ERR_NO_MEMORY, "out of host memory (std::bad_alloc)");
    } catch (const std::exception &e) {
        return open_fail(KVM_ERR_DEVICE, e.what());
    } catch (...) {
        return open_fail(KVM_ERR_DEVICE, "unknown C++ exception");
    }
}
template <typename F>
int guarded(kvm_group *g, F &&body) noexcept {
    if (!g) return KVM_ERR_INVALID_HANDLE;
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return gfail(g, KVM_ERR_NO_MEMORY, "out of host memory (std::bad_alloc)");
    } catch (const std::exception &e) {
        return gfail(g, KVM_ERR_DEVICE, e.what());
    } catch (...) {
        return gfail(g, KVM_ERR_DEVICE, "unknown C++ exception");
    }
}
}  // namespace

extern "C" int kvm_open_devices(const int32_t *devices, int32_t n_dev, int32_t in_flight, int32_t flags, kvm_group **group) {
    return guarded_open([&] { return open_common(devices, n_dev, 0, n_dev, nullptr, true, in_flight, flags, group); });
}
extern "C" int kvm_group_unique_id(void *id, int32_t id_bytes) {
    return guarded_open([&] {
        const int rc = kvm_sd_unique_id(id, id_bytes);
        return rc ? open_fail(rc, kvm_sd_last_error()) : KVM_OK;
    });
}
extern "C" int kvm_open_rank(int32_t device, int32_t rank, int32_t world, const void *id, int32_t id_bytes, int32_t in_flight,
                             int32_t flags, kvm_group **group) {
    return guarded_open([&] {
        const bool rccl = world > 1 || (flags & KVM_GROUP_RCCL_ALWAYS);
        if (rccl && (!id || id_bytes != KVM_GROUP_ID_BYTES))
            return open_fail(KVM_ERR_ILLEGAL_ARGUMENT, "kvm_open_rank: id must be the KVM_GROUP_ID_BYTES bytes of kvm_group_unique_id");
        return open_common(&device, 1, rank, world, id, false, in_flight, flags, group);
    });
}
extern "C" int kvm_group_close(kvm_group *g) {
    if (!g) return KVM_ERR_INVALID_HANDLE;
    destroy_group(g);
    return KVM_OK;
}
extern "C" const char *kvm_group_last_error(const kvm_group *g) { return g ? g->err.c_str() : t_open_err.c_str(); }
extern "C" int32_t kvm_group_world(const kvm_group *g) { return g ? g->world : KVM_ERR_INVALID_HANDLE; }
extern "C" int32_t kvm_group_local_devices(const kvm_group *g) { return g ? (int32_t)g->devs.size() : KVM_ERR_INVALID_HANDLE; }
extern "C" int32_t kvm_group_rccl_ranks(const kvm_group *g) { return g ? (g->use_rccl ? g->world : 0) : KVM_ERR_INVALID_HANDLE; }
extern "C" int64_t kvm_group_stat(const kvm_group *g, int32_t which) {
    return (g && which >= 0 && which < KVM_GROUP_STAT_COUNT) ? g->stat[which].load() : 0;
}
extern "C" kvm_ctx *kvm_group_ctx(kvm_group *g, int32_t local_device, int32_t slot) {
    if (!g || local_device < 0 || local_device >= (int32_t)g->devs.size() || slot < 0 || slot >= g->in_flight) return nullptr;
    return g->devs[(size_t)local_device].slots[(size_t)slot].ctx;
}
extern "C" int kvm_group_configure(kvm_group *g, int32_t what, int64_t value) {
    return guarded(g, [&] {
        if (what == KVM_GROUP_SPLIT_MODE) {
            if (value < KVM_SPLIT_AUTO || value > KVM_SPLIT_CONTIGUOUS) return gfail(g, KVM_ERR_ILLEGAL_ARGUMENT, "unknown split mode");
            g->split_mode = (int)value;
            return KVM_OK;
        }
        if (what == KVM_GROUP_PACK_CAP) {
            if (value < 1 || value > (1 << 24)) return gfail(g, KVM_ERR_ILLEGAL_ARGUMENT, "pack_cap must be 1 .. 2^24");
            if (value == g->pack_cap) return KVM_OK;
            free_exchange(g);   // (no call is running: every call drains what it enqueued)
            g->pack_cap = value;
            return alloc_exchange(g);
        }
        return gfail(g, KVM_ERR_ILLEGAL_ARGUMENT, "unknown kvm_group_configure item");
    });
}

namespace {
template <typename Load>
int load_everywhere(kvm_group *g, Load &&load) {
    g->loaded = false;
    for (gdev_t &d : g->devs) {
        kvm_ctx *owner = d.slots[0].ctx;
        // (the slots that share the owner's arrays must let go of them before the owner frees them)
        int rc = load(owner);
        if (rc) return gfail(g, rc, kvm_last_error(owner));
        for (size_t t = 1; t < d.slots.size(); t++)
            if ((rc = kvm_share_series(d.slots[t].ctx, owner))) return gfail(g, rc, kvm_last_error(d.slots[t].ctx));
    }
    g->loaded = true;
    return KVM_OK;
}
}  // namespace
extern "C" int kvm_group_load_series_f64(kvm_group *g, const double *host_values, int64_t n) {
    return guarded(g, [&] { return load_everywhere(g, [&](kvm_ctx *c) { return kvm_load_series_f64(c, host_values, n); }); });
}
extern "C" int kvm_group_load_series_be64(kvm_group *g, const void *be_bytes, int64_t n) {
    return guarded(g, [&] { return load_everywhere(g, [&](kvm_ctx *c) { return kvm_load_series_be64(c, be_bytes, n); }); });
}
extern "C" int kvm_group_load_series_file(kvm_group *g, const char *path, int64_t n) {
    return guarded(g, [&] { return load_everywhere(g, [&](kvm_ctx *c) { return kvm_load_series_file(c, path, n); }); });
}
extern "C" int kvm_group_set_option(kvm_group *g, int32_t option, int64_t value) {
    return guarded(g, [&] {
        for (gdev_t &d : g->devs)
            for (gslot_t &s : d.slots)
                if (int rc = kvm_set_option(s.ctx, option, value)) return gfail(g, rc, kvm_last_error(s.ctx));
        return (int)KVM_OK;
    });
}

// ---- a stream of sharded queries ----------------------------------------------------------------------------------
//
// Protocol of one kvm_run_sharded_jobs call.  Job i runs on slot i mod F of every local device.
//   slot thread (d, t), for i = t, t + F, ...:  wait until job i - F has been handed back; split the list (the first thread to
//       reach the job does it for all local devices); enqueue kvm_verify_device + kvm_pack_answers_device on the slot's stream.
//       The thread that completes the job's LAST local part is the job's issuer: it waits for its turn -- the exchanges are
//       issued in job order, on every rank, whichever thread issues them -- and puts the job's exchange onto the slots' own
//       streams right behind their kernels: the all-gather (one RCCL group over the local devices), on local device 0 the
//       readback kernel into pinned host memory and the `issued` event.
//   calling thread, for i = 0, 1, ...:  wait until job i's exchange is issued, wait for its event, merge the blocks of all
//       ranks, sort, hand the slots back.
// What must be the same on every rank is the ORDER of collectives on the communicator.  The turnstile gives gather(0),
// gather(1), ...; two things interleave with it and are pinned to a position every rank computes alike:
//   * the exact-size second exchange of a job i whose block overflowed: issued by the calling thread once gather(H - 1) has been
//     issued, H = min(i + F, n_jobs) -- the jobs below H need no slot of job i or later, so their gathers are issued without
//     anybody's help, and gather(i + F) cannot be issued before job i is handed back;
//   * the end of a stream at a failing job i (every rank sees the failure mark in the gathered blocks): every rank still issues
//     the gathers below the same H, then stops -- none is left with a collective its peers never join.
namespace {
struct job_state {
    std::mutex split_mu;   // the first slot thread to reach the job splits its list for every local device
    bool split_done = false;
    int split_rc = KVM_OK;
    std::string split_msg;
    std::vector<std::vector<int32_t>> L, R;  // per local device
    int64_t cnt_candidate = 0;               // sum(right - left + 1) over the WHOLE list, as the unsharded call counts it
    std::chrono::steady_clock::time_point t0;
    // guarded by run_state::mu
    std::vector<int> rc;           // per local device
    std::vector<std::string> msg;
    int produced = 0;              // local devices whose part of the job is enqueued
    bool issued = false;           // the job's exchange is on the streams (or could not be put there: issue_rc)
    int issue_rc = KVM_OK;
    std::string issue_msg;
    bool consumed = false;         // merged: the slots of the job are free again
};
struct run_state {
    std::mutex mu;
    std::condition_variable cv_issued;                 // the calling thread waits here (a job issued, the turnstile moved)
    std::condition_variable cv_turn;                   // issuers wait here for their turn
    std::vector<std::condition_variable> cv_consumed;  // the threads of slot index t wait on [t]: a job handed back wakes only
    bool stop = false;                                 // its own slots' threads, not all in_flight x devices of them
    int64_t next_issue = 0;                            // the turnstile: the job whose exchange is issued next
    std::vector<job_state> js;
    run_state(size_t n, size_t slots) : cv_consumed(slots), js(n) {}
    void end() {   // every loop leaves at its next wait
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        for (auto &cv : cv_consumed) cv.notify_all();
        cv_turn.notify_all();
    }
};

int reserve_slot(const gdev_t &d, gslot_t &s, int64_t cap) {
    if (cap <= s.cap) return KVM_OK;
    // (the slot is idle: its previous job has been merged, every kernel that read these arrays has finished)
    kvm_sd_free(d.device, s.d_off);
    kvm_sd_free(d.device, s.d_dist);
    s.d_off = nullptr;
    s.d_dist = nullptr;
    s.cap = 0;
    const int64_t want = std::max<int64_t>(cap + cap / 8, 4096);
    int rc = kvm_sd_malloc(d.device, (void **)&s.d_off, (size_t)want * 4);
    if (!rc) rc = kvm_sd_malloc(d.device, (void **)&s.d_dist, (size_t)want * 8);
    if (rc) return rc;
    s.cap = want;
    return KVM_OK;
}

void split_job(const kvm_group *g, const kvm_job &j, job_state &J) {
    J.t0 = std::chrono::steady_clock::now();
    const int nl = (int)g->devs.size();
    try {
        if (j.k < 0 || (j.k > 0 && (!j.left || !j.right))) {
            J.split_rc = KVM_ERR_ILLEGAL_ARGUMENT;
            J.split_msg = "bad interval list (k < 0 or null arrays)";
        } else {
            int64_t cnt = 0;
            for (int32_t i = 0; i < j.k; i++) cnt += (int64_t)j.right[i] - (int64_t)j.left[i] + 1;  // K/QueryEngine.java:344
            J.cnt_candidate = cnt;
            split_parts(j.left, j.right, j.k, g->world, g->split_mode, g->rank0, nl, J.L, J.R);
        }
    } catch (...) {
        J.split_rc = KVM_ERR_NO_MEMORY;
        J.L.clear();
        J.R.clear();
        try {
            J.split_msg = "out of host memory while splitting the interval list";
        } catch (...) {
        }
    }
    J.split_done = true;
}

int64_t lap_ns(std::chrono::steady_clock::time_point &t0) {
    const auto now = std::chrono::steady_clock::now();
    const int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(now - t0).count();
    t0 = now;
    return ns;
}

// Job i's exchange onto its slots' OWN streams, behind the kernels the slot threads enqueued there (no thread touches these
// streams again before the job is handed back): stream order is the only dependency.  Called in job order (the turnstile).
int issue_exchange(kvm_group *g, job_state &J, int64_t i, std::string &msg) {
    const int nl = (int)g->devs.size();
    const size_t t = (size_t)(i % g->in_flight), bd = 2 * ((size_t)g->pack_cap + 1);
    auto tp = std::chrono::steady_clock::now();
    std::vector<int> devices((size_t)nl);
    std::vector<void *> comms((size_t)nl), recv((size_t)nl), streams((size_t)nl);
    std::vector<const void *> send((size_t)nl);
    for (int d = 0; d < nl; d++) {
        gdev_t &dv = g->devs[(size_t)d];
        gslot_t &s = dv.slots[t];
        // (a part that failed has marked its block; once more here, in case THAT is what failed)
        if (J.rc[(size_t)d]) (void)kvm_poison_packed_device(s.ctx, s.d_packed, s.stream);
        devices[(size_t)d] = dv.device;
        comms[(size_t)d] = dv.comm;
        send[(size_t)d] = s.d_packed;
        recv[(size_t)d] = s.d_gathered;
        streams[(size_t)d] = s.stream;
    }
    gdev_t &d0 = g->devs[0];
    void *s0 = d0.slots[t].stream;
    g->stat[KVM_GROUP_STAT_PREPARE_NS] += lap_ns(tp);
    int rc = KVM_OK;
    if (g->use_rccl) {
        // Consecutive jobs run on different streams, and so do their all-gathers.  With peers, two collectives of one
        // communicator must not be in flight at once (they share its channels): each device's gather is ordered behind its
        // previous one by an event -- whatever the collective library does about streams on its own.  (At world 1 there is
        // nobody to be out of step with, and the wait would only put barrier packets into the slots' hardware queues.)
        // (also a call's first job: on a local device other than 0 the last gather of the previous call may still be finishing --
        // calls wait for device 0's readbacks; an event that was never recorded counts as complete)
        if (g->world > 1)
            for (int d = 0; d < nl && !rc; d++) rc = kvm_sd_stream_wait_event(g->devs[(size_t)d].device, streams[(size_t)d], g->devs[(size_t)d].gathered);
        if (!rc) rc = kvm_sd_all_gather(nl, devices.data(), comms.data(), send.data(), recv.data(), bd, streams.data());
        if (g->world > 1)
            for (int d = 0; d < nl && !rc; d++) rc = kvm_sd_event_record(g->devs[(size_t)d].device, g->devs[(size_t)d].gathered, streams[(size_t)d]);
        g->stat[KVM_GROUP_STAT_ALL_GATHER_NS] += lap_ns(tp);
        if (!rc) rc = kvm_sd_readback_blocks(d0.device, g->h_block[t], d0.slots[t].d_gathered, g->world, bd, s0);
    } else {
        rc = kvm_sd_readback_blocks(d0.device, g->h_block[t], d0.slots[t].d_packed, 1, bd, s0);
    }
    if (!rc) rc = kvm_sd_event_record(d0.device, g->issued[t], s0);
    g->stat[KVM_GROUP_STAT_COPY_NS] += lap_ns(tp);
    if (rc) msg = kvm_sd_last_error();
    return rc;
}

// The loop of slot t of local device d: jobs t, t + F, t + 2F ...
void slot_loop(kvm_group *g, run_state &rs, const kvm_job *jobs, int64_t n_jobs, int d, int t) {
    gdev_t &dv = g->devs[(size_t)d];
    gslot_t &s = dv.slots[(size_t)t];
    const int F = g->in_flight, nl = (int)g->devs.size();
    for (int64_t i = t; i < n_jobs; i += F) {
        auto tp = std::chrono::steady_clock::now();
        {
            std::unique_lock<std::mutex> lk(rs.mu);
            rs.cv_consumed[(size_t)t].wait(lk, [&] { return rs.stop || i - F < 0 || rs.js[(size_t)(i - F)].consumed; });
            if (rs.stop) return;
        }
        g->stat[KVM_GROUP_STAT_SLOT_WAIT_NS] += lap_ns(tp);
        job_state &J = rs.js[(size_t)i];
        {
            std::lock_guard<std::mutex> sl(J.split_mu);
            if (!J.split_done) split_job(g, jobs[i], J);
        }
        g->stat[KVM_GROUP_STAT_SLOT_SPLIT_NS] += lap_ns(tp);
        int rc = J.split_rc;
        std::string msg;
        try {
            if (rc) {
                msg = J.split_msg;
            } else {
                const kvm_job &j = jobs[i];
                const std::vector<int32_t> &L = J.L[(size_t)d], &R = J.R[(size_t)d];
                int64_t cands = 0;
                for (size_t q = 0; q < L.size(); q++) cands += (int64_t)R[q] - L[q] + 1;
                if ((rc = reserve_slot(dv, s, cands + 1))) {
                    msg = kvm_sd_last_error();
                } else {
                    static const int32_t none = 0;  // (an empty shard still needs non-null arrays)
                    int64_t cnt = 0;
                    kvm_host::first_call_alone guard;
                    rc = kvm_verify_device(s.ctx, j.variant, j.query, j.m, L.empty() ? &none : L.data(), R.empty() ? &none : R.data(),
                                           (int32_t)L.size(), j.shift, j.epsilon, j.rho, j.alpha, j.beta, s.d_off, s.d_dist, s.cap,
                                           s.d_count, &cnt, s.stream);
                    if (rc == KVM_OK) rc = kvm_pack_answers_device(s.ctx, s.d_off, s.d_dist, s.d_count, g->pack_cap, s.d_packed, s.stream);
                    if (rc) msg = kvm_last_error(s.ctx);
                }
            }
        } catch (...) {
            rc = KVM_ERR_NO_MEMORY;
        }
        if (rc) (void)kvm_poison_packed_device(s.ctx, s.d_packed, s.stream);  // the failure mark every rank will see
        g->stat[KVM_GROUP_STAT_SLOT_ENQUEUE_NS] += lap_ns(tp);
        bool issuer;
        {
            std::unique_lock<std::mutex> lk(rs.mu);
            J.rc[(size_t)d] = rc;
            J.msg[(size_t)d].swap(msg);
            issuer = ++J.produced == nl;
            if (issuer) {
                rs.cv_turn.wait(lk, [&] { return rs.stop || rs.next_issue == i; });
                if (rs.stop) return;
            }
        }
        if (!issuer) continue;
        g->stat[KVM_GROUP_STAT_SLOT_TURN_NS] += lap_ns(tp);
        std::string imsg;
        int irc;
        try {
            irc = issue_exchange(g, J, i, imsg);
        } catch (...) {
            irc = KVM_ERR_NO_MEMORY;
        }
        {
            std::lock_guard<std::mutex> lk(rs.mu);
            J.issue_rc = irc;
            J.issue_msg.swap(imsg);
            J.issued = true;
            rs.next_issue = i + 1;
        }
        rs.cv_turn.notify_all();
        rs.cv_issued.notify_all();
    }
}

struct exchange {
    kvm_group *g;
    run_state &rs;
    int64_t n_jobs;
    kvm_result **results;
    int nl;
    size_t bd;  // doubles per rank block

    // every gather below H = min(i + F, n_jobs) is on the streams (see the protocol note above); false: the stream was stopped
    void wait_horizon(int64_t i) {
        const int64_t H = std::min<int64_t>(i + g->in_flight, n_jobs);
        std::unique_lock<std::mutex> lk(rs.mu);
        rs.cv_issued.wait(lk, [&] { return rs.next_issue >= H; });
    }

    // more answers on some rank than the fixed-size block holds (rare): a second, exact-size exchange of the slots' full lists
    int exchange_exact(int64_t i, int64_t mx, std::vector<double> &host) {
        wait_horizon(i);
        const size_t t = (size_t)(i % g->in_flight);
        const size_t big = 2 * ((size_t)mx + 1);
        std::vector<void *> d_big((size_t)nl, nullptr), d_all((size_t)nl, nullptr);
        double *h = nullptr;
        int rc = KVM_OK;
        std::vector<int> devices((size_t)nl);
        std::vector<void *> comms((size_t)nl), streams((size_t)nl);
        std::vector<const void *> send((size_t)nl);
        for (int d = 0; d < nl && rc == KVM_OK; d++) {
            gdev_t &dv = g->devs[(size_t)d];
            gslot_t &s = dv.slots[t];
            rc = kvm_sd_malloc(dv.device, &d_big[(size_t)d], big * 8);
            if (!rc && g->use_rccl) rc = kvm_sd_malloc(dv.device, &d_all[(size_t)d], big * 8 * (size_t)g->world);
            if (rc) {
                (void)gfail_sd(g, rc);
                break;
            }
            // (on the slot's stream, behind its kernels and the first exchange)
            rc = kvm_pack_answers_device(s.ctx, s.d_off, s.d_dist, s.d_count, mx, (double *)d_big[(size_t)d], s.stream);
            if (rc) (void)gfail(g, rc, kvm_last_error(s.ctx));
            devices[(size_t)d] = dv.device;
            comms[(size_t)d] = dv.comm;
            send[(size_t)d] = d_big[(size_t)d];
            streams[(size_t)d] = s.stream;
        }
        gdev_t &d0 = g->devs[0];
        const size_t total = big * (size_t)g->world;
        if (!rc && (rc = kvm_sd_host_alloc(d0.device, (void **)&h, total * 8))) (void)gfail_sd(g, rc);
        if (!rc && g->use_rccl) {
            // (the turnstile is at H and stays there until this job is handed back: no gather is issued beside this collective)
            // (ordered behind the last gather like every gather behind its predecessor; the streams are drained below, so the next
            // gather needs no event of this one)
            if (g->world > 1)
                for (int d = 0; d < nl && !rc; d++)
                    if ((rc = kvm_sd_stream_wait_event(g->devs[(size_t)d].device, streams[(size_t)d], g->devs[(size_t)d].gathered))) (void)gfail_sd(g, rc);
            if (!rc && (rc = kvm_sd_all_gather(nl, devices.data(), comms.data(), send.data(), d_all.data(), big, streams.data()))) (void)gfail_sd(g, rc);
        }
        if (!rc && (rc = kvm_sd_copy_d2h_async(d0.device, h, g->use_rccl ? d_all[0] : d_big[0], total * 8, d0.slots[t].stream))) (void)gfail_sd(g, rc);
        for (int d = 0; d < nl; d++) {  // (also on a failure: the buffers below are freed next)
            const int rs_ = kvm_sd_stream_synchronize(g->devs[(size_t)d].device, g->devs[(size_t)d].slots[t].stream);
            if (rs_ && !rc) rc = gfail_sd(g, rs_);
        }
        if (!rc) {
            try {
                host.assign(h, h + total);
            } catch (...) {
                rc = gfail(g, KVM_ERR_NO_MEMORY, "out of host memory (exact-size exchange)");
            }
        }
        kvm_sd_host_free(h);
        for (int d = 0; d < nl; d++) {
            kvm_sd_free(g->devs[(size_t)d].device, d_big[(size_t)d]);
            kvm_sd_free(g->devs[(size_t)d].device, d_all[(size_t)d]);
        }
        return rc;
    }

    // wait for job i's exchange and its block, merge it into results[i].  Returns the job's status; *exchange_broken: the
    // exchange itself failed on this rank (nothing sensible is left to do), as opposed to a shard that failed somewhere.
    int finish(int64_t i, bool *exchange_broken) {
        job_state &J = rs.js[(size_t)i];
        const size_t t = (size_t)(i % g->in_flight);
        auto tp = std::chrono::steady_clock::now();
        {
            std::unique_lock<std::mutex> lk(rs.mu);
            rs.cv_issued.wait(lk, [&] { return J.issued; });
        }
        g->stat[KVM_GROUP_STAT_WAIT_ISSUED_NS] += lap_ns(tp);
        if (J.issue_rc) {
            *exchange_broken = true;
            return gfail(g, J.issue_rc, J.issue_msg);
        }
        if (int rc = kvm_sd_event_synchronize(g->issued[t])) {
            *exchange_broken = true;
            return gfail_sd(g, rc);
        }
        g->stat[KVM_GROUP_STAT_WAIT_BLOCK_NS] += lap_ns(tp);
        const double *h = g->h_block[t];
        const int world = g->world;
        int64_t mx = 0, total = 0;
        std::string bad;
        for (int w = 0; w < world; w++) {
            const double c = h[(size_t)w * bd];
            if (c < 0) bad += (bad.empty() ? "" : ", ") + std::to_string(w);
            mx = std::max<int64_t>(mx, (int64_t)c);
            total += c > 0 ? (int64_t)c : 0;
        }
        for (int d = 0; d < nl; d++)   // this process's own failure: its status and text are what the call returns
            if (J.rc[(size_t)d]) return gfail(g, J.rc[(size_t)d], J.msg[(size_t)d]);
        if (!bad.empty()) return gfail(g, KVM_ERR_SHARD_FAILED, "rank(s) " + bad + " failed on their shard of job " + std::to_string(i));
        std::vector<int32_t> off;
        std::vector<double> dist;
        off.reserve((size_t)total);
        dist.reserve((size_t)total);
        std::vector<double> exact;
        size_t stride = bd;
        if (mx > g->pack_cap) {
            if (int rc = exchange_exact(i, mx, exact)) {
                *exchange_broken = true;
                return rc;
            }
            h = exact.data();
            stride = 2 * ((size_t)mx + 1);
        }
        for (int w = 0; w < world; w++) {
            const double *b = h + (size_t)w * stride;
            const int64_t c = (int64_t)b[0];
            for (int64_t q = 1; q <= c; q++) {
                off.push_back((int32_t)b[2 * q]);  // (an int32 offset is exact in fp64)
                dist.push_back(b[2 * q + 1]);
            }
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - J.t0).count();
        kvm_result *r = kvm_result_from_pairs(off.data(), dist.data(), (int64_t)off.size(), J.cnt_candidate, ms);
        if (!r) return gfail(g, KVM_ERR_NO_MEMORY, "out of host memory (result)");
        results[i] = r;
        g->stat[KVM_GROUP_STAT_MERGE_NS] += lap_ns(tp);
        return KVM_OK;
    }
};

int run_sharded(kvm_group *g, const kvm_job *jobs, int64_t n_jobs, kvm_result **results, int32_t *status) {
    if (n_jobs < 0 || (n_jobs > 0 && (!jobs || !results))) return gfail(g, KVM_ERR_ILLEGAL_ARGUMENT, "bad job list");
    for (int64_t i = 0; i < n_jobs; i++) {
        results[i] = nullptr;
        if (status) status[i] = KVM_OK;
    }
    if (n_jobs == 0) return KVM_OK;
    if (!g->loaded) return gfail(g, KVM_ERR_NO_SERIES, "no series loaded (kvm_group_load_series_*)");
    const int nl = (int)g->devs.size(), F = g->in_flight;
    const int used = (int)std::min<int64_t>(F, n_jobs);
    for (auto &c : g->stat) c = 0;
    g->stat[KVM_GROUP_STAT_JOBS] = n_jobs;
    run_state rs((size_t)n_jobs, (size_t)F);
    for (job_state &J : rs.js) {
        J.rc.assign((size_t)nl, KVM_OK);
        J.msg.resize((size_t)nl);
    }
    exchange ex{g, rs, n_jobs, results, nl, block_doubles(g)};
    int first_rc = KVM_OK;
    {
        auto get = [&](int32_t q) { return g->devs[(size_t)(q / F)].slots[(size_t)(q % F)].ctx; };
        kvm_host::in_flight_mark<decltype(get)> mark(get, nl * F, used);
        kvm_host::thread_group tg;  // (waits for every loop on every way out, before rs and the job arrays go away)
        // On EVERY way out of this block -- the stream's end, a failing job, an exception (a worker that could not be started,
        // std::bad_alloc in the merge) -- the loops are told to stop before ~thread_group waits for them.
        struct stopper {
            run_state &rs;
            ~stopper() { rs.end(); }
        } stop_on_exit{rs};
        for (int d = 0; d < nl; d++)
            for (int t = 0; t < used; t++) tg.start([g, &rs, jobs, n_jobs, d, t] { slot_loop(g, rs, jobs, n_jobs, d, t); });
        for (int64_t i = 0; i < n_jobs; i++) {
            bool broken = false;
            const int rc = ex.finish(i, &broken);
            if (status) status[i] = rc;
            if (rc) {
                first_rc = rc;
                // a shard failed (here or on another rank): every rank sees it at this job; each still issues the gathers its
                // peers are going to issue (see the protocol note), then the stream ends
                if (!broken) ex.wait_horizon(i);
                break;
            }
            {
                std::lock_guard<std::mutex> lk(rs.mu);
                rs.js[(size_t)i].consumed = true;
                rs.js[(size_t)i].L.clear();
                rs.js[(size_t)i].R.clear();
            }
            rs.cv_consumed[(size_t)(i % F)].notify_all();
        }
    }
    // an early end: the kernels of the jobs that ran ahead still read the contexts' scratch, and the gathers already issued
    // (every rank issued the same ones) must have completed before the buffers are used again
    if (first_rc != KVM_OK)
        for (gdev_t &d : g->devs)
            for (int t = 0; t < used; t++) (void)kvm_sd_stream_synchronize(d.device, d.slots[(size_t)t].stream);
    return first_rc;
}
}  // namespace

extern "C" int kvm_run_sharded_jobs(kvm_group *g, const kvm_job *jobs, int64_t n_jobs, kvm_result **results, int32_t *status) {
    return guarded(g, [&] { return run_sharded(g, jobs, n_jobs, results, status); });
}

extern "C" int kvm_verify_sharded(kvm_group *g, int32_t variant, const double *query, int32_t m, const int32_t *left,
                                  const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho, double alpha,
                                  double beta, kvm_result **result) {
    return guarded(g, [&] {
        if (!result) return gfail(g, KVM_ERR_ILLEGAL_ARGUMENT, "null result pointer");
        *result = nullptr;
        kvm_job j;
        std::memset(&j, 0, sizeof j);
        j.variant = variant;
        j.m = m;
        j.query = query;
        j.left = left;
        j.right = right;
        j.k = k;
        j.shift = shift;
        j.epsilon = epsilon;
        j.rho = rho;
        j.alpha = alpha;
        j.beta = beta;
        return run_sharded(g, &j, 1, result, nullptr);
    });
}
