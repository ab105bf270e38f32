// kvm_index_file.cpp -- host side of the index build: the part of IndexBuilder.run behind step 1
// (K/IndexBuilder.java:296-347) and the bytes of the index file (K/operator/file/IndexFileOperator.java:127-165).
//
// Step 1 -- the per-position row keys and the intervals of equal keys, the part that touches every point -- runs on the
// GPU (kvm_index.hip, kvm_index_rows_build_grouped) and hands over the reference's rows (one IndexNode per distinct key,
// intervals in append order) in the order step 2 walks them.  What follows is int / byte work over those intervals:
//   step 2   walk the rows by descending key; a row with fewer intervals than 1.2 x the average is merged into the
//            running row (IndexNodeUtils.mergeIndexNode, K/utils/IndexNodeUtils.java:30-90: union of two sorted interval
//            lists, overlapping or adjacent intervals joined, every result cut to <= 256 positions) if the union has
//            fewer than 0.8 x the intervals of the two; a finished row is stored under the key of the LAST row merged
//            into it, i.e. the lowest (:323-347);
//   codec    IndexNode.toBytesCompact (K/common/entity/IndexNode.java:61-100): groups of up to 255 intervals as
//            {left:int32 BE}{count-1}{right-left}{gap}{right-left}..., every small number biased by -128;
//   file     rows in ascending key order, each {key:double BE}{compact bytes}; the statistic triples {key, running
//            #intervals, running #offsets} (ByteUtils.listTripleToByteArray); the table of line offsets, whose last entry
//            is its own position (IndexFileOperator.writeAll / readOffsetInfo).  Offsets are Java ints: an image of 2 GiB
//            or more is refused.
// The walk is sequential (each decision depends on the row merged so far) except at the rows the rule never merges, where it
// starts afresh: the row sequence is cut there and walked by several host threads (see kvm_index_build); inside the long
// chains of sparse rows at both ends of the key range two threads work as a pipeline -- the next row's union is merged while
// this row's is still being written (pipe_rec); one pass per attempted union, read straight from the grouped step-1 arrays; the
// lines of the file are encoded by the same number of host threads.
#include "kvm_guard.h"
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kvmatch.h"

void kvm_set_error(kvm_ctx *ctx, const std::string &msg);  // kvm_api.hip
int kvm_index_rows_build_for_file(kvm_ctx *ctx, int32_t w, kvm_index_rows **out);  // kvm_api_index.hip: grouped rows, group keys only

// The image of a file that has been freed is kept for the next build (one buffer, the largest seen): a process builds the five
// indexes of a series one after the other, and 100 MB of fresh pages cost the encoder's threads ~10 ms of faults per call.
namespace {
struct image_pool_t {
    std::mutex mu;
    std::unique_ptr<uint8_t[]> p;
    size_t cap = 0;
};
image_pool_t *image_pool() {
    static image_pool_t *pool = new image_pool_t();   // (never destroyed: a file may be freed at exit)
    return pool;
}
}  // namespace
struct kvm_index_file {
    std::unique_ptr<uint8_t[]> image;   // (not a vector: its threads first-touch their own parts; nothing zero-fills 100 MB)
    size_t image_size = 0, image_cap = 0;
    void take_image(size_t bytes) {
        image_pool_t *pool = image_pool();
        {
            std::lock_guard<std::mutex> lk(pool->mu);
            if (pool->cap >= bytes) {
                image = std::move(pool->p);
                image_cap = pool->cap;
                pool->cap = 0;
            }
        }
        if (!image) {
            image.reset(new uint8_t[bytes]);
            image_cap = bytes;
        }
        image_size = bytes;
    }
    ~kvm_index_file() {
        if (!image) return;
        image_pool_t *pool = image_pool();
        std::lock_guard<std::mutex> lk(pool->mu);
        if (image_cap > pool->cap) {
            pool->p = std::move(image);
            pool->cap = image_cap;
        }
    }
    int64_t rows_before = 0, rows_after = 0, intervals_before = 0, intervals_after = 0;
    double step1_ms = 0, step2_ms = 0, encode_ms = 0;
};

namespace {

struct ival {
    int32_t first, last;
};

constexpr int32_t MAXIMUM_DIFF = 256;  // IndexNode.MAXIMUM_DIFF

// A row's positions: either a slice of the grouped step-1 arrays or a vector of merged intervals
struct raw_view {
    const int32_t *f, *l;
    int64_t n;
    int64_t size() const { return n; }
    ival at(int64_t i) const { return ival{f[i], l[i]}; }
};
struct vec_view {
    const ival *p;
    int64_t n;
    int64_t size() const { return n; }
    ival at(int64_t i) const { return p[i]; }
};

// IndexNodeUtils.mergeIndexNode; Emit(first, last) receives every interval handed to addInterval.
// Bulk(view, i, j) receives a RUN of intervals [i, j) of one list that the walk would hand over one by one unchanged -- every one
// of them ends more than one position before the interval the other list is holding (`l.last + 1 < other.first`: emitted as it
// is, the held interval untouched).  The run's end is found by galloping (1, 2, 4 ... then bisection): where a short row is
// merged into a long union -- the chains of sparse rows at both ends of the key range, the sequential part of step 2 -- most of
// the long list goes through in runs, copied instead of compared one by one.  Same intervals, same order.
template <class N>
inline int64_t run_end(const N &n, int64_t i, int64_t size, int32_t other_first) {
    // the first index >= i whose interval does NOT end more than one position before other_first (ends are ascending)
    int64_t step = 1, lo = i, hi = i + 1;
    while (hi < size && (int64_t)n.at(hi).last + 1 < other_first) {
        lo = hi;
        step *= 2;
        hi = std::min(size, hi + step);
    }
    // n[lo] qualifies; n[hi] does not (or hi == size)
    while (lo + 1 < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if ((int64_t)n.at(mid).last + 1 < other_first) lo = mid;
        else hi = mid;
    }
    return hi;
}
template <class N1, class N2, class Emit, class Bulk1, class Bulk2>
inline void merge_walk(const N1 &n1, const N2 &n2, Emit emit, Bulk1 bulk1, Bulk2 bulk2) {
    const int64_t s1 = n1.size(), s2 = n2.size();
    int64_t i1 = 0, i2 = 0;
    bool h1 = false, h2 = false;
    ival l1{0, 0}, l2{0, 0};
    while (i1 < s1 && i2 < s2) {
        if (!h1) { l1 = n1.at(i1); h1 = true; }
        if (!h2) { l2 = n2.at(i2); h2 = true; }
        if (l1.last + 1 < l2.first) {
            emit(l1.first, l1.last);
            i1++;
            h1 = false;
            // (the intervals behind it are fresh ones of list 1: a run of them that also ends before l2 goes over as it is)
            if (i1 + 1 < s1 && (int64_t)n1.at(i1 + 1).last + 1 < l2.first) {
                const int64_t j = run_end(n1, i1, s1, l2.first);
                bulk1(n1, i1, j);
                i1 = j;
            }
        } else if (l2.last + 1 < l1.first) {
            emit(l2.first, l2.last);
            i2++;
            h2 = false;
            if (i2 + 1 < s2 && (int64_t)n2.at(i2 + 1).last + 1 < l1.first) {
                const int64_t j = run_end(n2, i2, s2, l1.first);
                bulk2(n2, i2, j);
                i2 = j;
            }
        } else if (l1.last < l2.last) {
            if (l1.first < l2.first) l2.first = l1.first;
            i1++;
            h1 = false;
        } else {
            if (l2.first < l1.first) l1.first = l2.first;
            i2++;
            h2 = false;
        }
    }
    for (int64_t i = i1; i < s1; i++) {
        if (!h1) l1 = n1.at(i);
        emit(l1.first, l1.last);
        h1 = false;
    }
    for (int64_t i = i2; i < s2; i++) {
        if (!h2) l2 = n2.at(i);
        emit(l2.first, l2.last);
        h2 = false;
    }
}

// ---- step 2 as a PIPELINE (the chains of sparse rows at both ends of the key range) ------------------------------------------
// There the walk is one long dependency: row i is merged into the union of the rows before it, 190 times in a row, each union
// ~4 500 intervals, and whether it is kept is known only at its end -- 22 of step 2's 37 million interval-steps sit in one
// range no cut can divide (kvm_index_build).  But a union is produced front to back and consumed front to back: the task of row
// i + 1 can merge into the union of row i WHILE that is being written, a few hundred intervals behind it, on the guess that
// the union will be kept (it is, 17 times of 18).  A team of D threads takes the rows of a range in turn (row i: thread
// i mod D); behind every row stands a record: the union's intervals published so far, and -- once the row's task has decided
// -- whether the walk goes on from that union or afresh from the row itself.  A task whose input turns out to be something
// else (the predecessor started again, or did not keep its union) starts again.  Same merges, same decisions, same order of
// finished rows as one thread's walk: the finishing of rows is handed from task to task with the decisions.
struct alignas(128) pipe_rec {   // (a record to itself on its cache lines: its neighbours are written by other threads)
    std::atomic<uint64_t> prog{0};   // (attempt << 40) | intervals of `buf` published by that attempt of the row's task
    std::atomic<int> decided{0};     // 0 not yet | 1 the walk goes on from the union in buf[0, n) | 2 from the row itself (n = its size)
    int64_t n = 0;
    int64_t first_row = 0;           // the first row of the chain the walk is in behind this row
    std::atomic<uint64_t> *buf = nullptr;   // packed intervals (relaxed atomics: a reader may look at an attempt being overwritten;
};                                          // it finds out at its next look at `prog` and throws away what it made of them)
constexpr uint64_t PIPE_N_MASK = (1ULL << 40) - 1;
inline uint64_t pipe_pack(ival v) { return (uint64_t)(uint32_t)v.first | ((uint64_t)(uint32_t)v.last << 32); }
inline ival pipe_unpack(uint64_t u) { return ival{(int32_t)(uint32_t)u, (int32_t)(uint32_t)(u >> 32)}; }
inline void pipe_relax(int &spins) {
    if (++spins > 256) std::this_thread::yield();
}
constexpr int64_t PIPE_BATCH = 256;   // intervals between two publications: the reader streams whole batches (a reader at the writer's
                                      // heels pulls every cache line across the chip one miss at a time)
struct pipe_sink {   // the output of a task's attempt: addInterval's cut, progress published every PIPE_BATCH intervals
    pipe_rec *r;
    int64_t cap, o = 0;
    uint64_t attempt;
    bool put(int64_t a, int64_t b) {
        while (b - a >= MAXIMUM_DIFF) {
            if (o >= cap) return false;
            r->buf[o++].store(pipe_pack(ival{(int32_t)a, (int32_t)(a + MAXIMUM_DIFF - 1)}), std::memory_order_relaxed);
            a += MAXIMUM_DIFF;
        }
        if (o >= cap) return false;
        r->buf[o++].store(pipe_pack(ival{(int32_t)a, (int32_t)b}), std::memory_order_relaxed);
        if ((o & (PIPE_BATCH - 1)) == 0) r->prog.store((attempt << 40) | (uint64_t)o, std::memory_order_release);
        return true;
    }
};
struct pipe_src_stream {   // the predecessor's union, as far as it has been published
    const pipe_rec *r;
    uint64_t attempt;
    int64_t seen = 0;
    // 1: v = interval i; 0: the union is complete and has no interval i; -1: this is not (any more) what the walk goes on from
    int get(int64_t i, ival &v) {
        int spins = 0;
        while (i >= seen) {
            const int d = r->decided.load(std::memory_order_acquire);
            const uint64_t p = r->prog.load(std::memory_order_acquire);
            if ((p >> 40) != attempt || d == 2) return -1;
            seen = (int64_t)(p & PIPE_N_MASK);
            if (i < seen) break;
            if (d == 1) return 0;   // (decided before the last look at prog: `seen` is the union's final size)
            pipe_relax(spins);
        }
        v = pipe_unpack(r->buf[i].load(std::memory_order_relaxed));
        return 1;
    }
};
struct pipe_src_raw {
    raw_view v;
    int get(int64_t i, ival &out) const {
        if (i >= v.n) return 0;
        out = v.at(i);
        return 1;
    }
};
// IndexNodeUtils.mergeIndexNode once more (merge_walk above, without the runs): list 1 from a source that may still be growing.
// 0: done; -1: the source is not what it seemed (start again); -2: the sink is full (a guess gone far wrong: wait, start again)
template <class Src>
inline int merge_from(Src &src, const raw_view &n2, pipe_sink &out) {
    const int64_t s2 = n2.n;
    int64_t i1 = 0, i2 = 0;
    bool h1 = false, h2 = false;
    ival l1{0, 0}, l2{0, 0};
    for (;;) {
        if (!h1) {
            const int g = src.get(i1, l1);
            if (g < 0) return -1;
            if (g == 0) break;
            h1 = true;
        }
        if (i2 >= s2) break;
        if (!h2) { l2 = n2.at(i2); h2 = true; }
        if ((int64_t)l1.last + 1 < (int64_t)l2.first) {
            if (!out.put(l1.first, l1.last)) return -2;
            i1++;
            h1 = false;
        } else if ((int64_t)l2.last + 1 < (int64_t)l1.first) {
            if (!out.put(l2.first, l2.last)) return -2;
            i2++;
            h2 = false;
        } else if (l1.last < l2.last) {
            if (l1.first < l2.first) l2.first = l1.first;
            i1++;
            h1 = false;
        } else {
            if (l2.first < l1.first) l1.first = l2.first;
            i2++;
            h2 = false;
        }
    }
    for (;;) {
        if (!h1) {
            const int g = src.get(i1, l1);
            if (g < 0) return -1;
            if (g == 0) break;
        }
        if (!out.put(l1.first, l1.last)) return -2;
        h1 = false;
        i1++;
    }
    for (; i2 < s2; i2++) {
        if (!h2) l2 = n2.at(i2);
        if (!out.put(l2.first, l2.last)) return -2;
        h2 = false;
    }
    return 0;
}

inline void put_be32(uint8_t *p, int32_t v) {
    const uint32_t u = (uint32_t)v;
    p[0] = (uint8_t)(u >> 24); p[1] = (uint8_t)(u >> 16); p[2] = (uint8_t)(u >> 8); p[3] = (uint8_t)u;
}
inline void put_be_double(uint8_t *p, double v) {
    uint64_t u;
    std::memcpy(&u, &v, 8);
    for (int b = 0; b < 8; b++) p[b] = (uint8_t)(u >> (56 - 8 * b));
}

// IndexNode.toBytesCompact into `result` (nullptr: only the length): groups of up to 255 intervals as
// {left:int32 BE}{count-1}{right-left}{gap}{right-left}..., every small number biased by -128.  Returns the length.
template <bool WRITE>
int64_t encode_compact(const ival *pos, int64_t size, uint8_t *result) {
    int64_t index = 0, length = 0;
    int count = 0;
    bool packing = false;
    while (index < size) {
        if (!packing) {
            if (WRITE) put_be32(result + length, pos[index].first);
            length += 4 + 1;
            if (WRITE) result[length] = (uint8_t)(pos[index].last - pos[index].first - 128);
            length++;
            packing = true;
            count = 1;
        } else {
            const int32_t gap = pos[index].first - pos[index - 1].last;
            if (gap < MAXIMUM_DIFF && (count - 1) / 2 + 2 < MAXIMUM_DIFF) {
                if (WRITE) {
                    result[length] = (uint8_t)(gap - 128);
                    result[length + 1] = (uint8_t)(pos[index].last - pos[index].first - 128);
                }
                length += 2;
                count += 2;
            } else {
                if (WRITE) result[length - count - 1] = (uint8_t)((count - 1) / 2 - 128);
                packing = false;
                continue;
            }
        }
        index++;
    }
    if (packing && WRITE) result[length - count - 1] = (uint8_t)((count - 1) / 2 - 128);
    return length;
}

// which ranges get a team: many rows, and rows long enough that a task's merge outweighs its hand-shakes (at n = 1e8 the sparse
// ends qualify at every window length of the reference's index set: 120 intervals per row at Wu = 400, 320 at Wu = 25).  The CPU
// harness lowers them (kvm_index_file_test_thresholds) so that lists of a few 10^5 points go through the pipeline under the
// sanitizers.
std::atomic<int64_t> g_team_min_rows{4096}, g_team_min_row_size{100}, g_team_size{0};   // (team size 0: by the number of CPUs)

// The CPUs that share a last-level cache with the one this thread runs on (/sys/devices/system/cpu/cpuN/cache/index3/
// shared_cpu_list, e.g. "8-15,136-143"), within what the thread may use: the two threads of a team hand every union from one to
// the other, and on a chip of many cache domains that costs three times as much across two of them.  false: unknown -- no pinning.
bool cache_mates(cpu_set_t *out) {
    const int cpu = sched_getcpu();
    if (cpu < 0) return false;
    char path[128];
    snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    FILE *fp = fopen(path, "r");
    if (!fp) return false;
    char buf[512];
    const bool got = fgets(buf, sizeof buf, fp) != nullptr;
    fclose(fp);
    if (!got) return false;
    cpu_set_t allowed;
    if (pthread_getaffinity_np(pthread_self(), sizeof allowed, &allowed) != 0) return false;
    CPU_ZERO(out);
    int n = 0;
    for (const char *q = buf; *q;) {
        char *end = nullptr;
        const long a = strtol(q, &end, 10);
        if (end == q) break;
        long b = a;
        q = end;
        if (*q == '-') {
            b = strtol(q + 1, &end, 10);
            q = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) {
                CPU_SET((int)c, out);
                n++;
            }
        while (*q == ',' || *q == ' ' || *q == '\n') q++;
    }
    return n >= 2;
}

double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

}  // namespace

void kvm_index_file_test_thresholds(int64_t rows, int64_t row_size, int64_t team_size) {   // (C++ linkage: not part of the ABI; tests/harness only)
    g_team_min_rows.store(rows);
    g_team_min_row_size.store(row_size);
    g_team_size.store(team_size);
}

extern "C" int kvm_index_build(kvm_ctx *ctx, int32_t w, kvm_index_file **out) {
    return kvm_guard<int>(ctx, [&]() -> int {
    if (!ctx) return KVM_ERR_INVALID_HANDLE;
    if (!out) {
        kvm_set_error(ctx, "null result pointer");
        return KVM_ERR_ILLEGAL_ARGUMENT;
    }
    *out = nullptr;
    const double t0 = now_ms();
    kvm_index_rows *rows = nullptr;
    int rc = kvm_index_rows_build_for_file(ctx, w, &rows);
    if (rc) return rc;
    const int64_t G = kvm_index_rows_groups(rows), cnt = kvm_index_rows_count(rows);
    if (G < 1) {  // (the Java code runs into rawStatisticInfo.get(0) on an empty list, K/IndexBuilder.java:321)
        kvm_index_rows_free(rows);
        kvm_set_error(ctx, "kvm_index_build: the series holds no window of that length");
        return KVM_ERR_ILLEGAL_ARGUMENT;
    }
    const double *gkey = kvm_index_rows_group_keys(rows);
    const int32_t *goff = kvm_index_rows_group_offsets(rows);
    const int32_t *rf = kvm_index_rows_first(rows), *rl = kvm_index_rows_last(rows);
    auto *f = new kvm_index_file();
    f->rows_before = G;
    f->intervals_before = cnt;
    const double t1 = now_ms();
    f->step1_ms = t1 - t0;

    // ---- step 2.  Group g is rawStatisticInfo.get(g) (descending key); its size is goff[g+1] - goff[g].
    // StatisticInfo.getAverage: the double sum of the sizes (exact: integers far below 2^53) over their number.
    const double average = (double)cnt / (double)G;
    // The walk is sequential: whether row i is merged depends on the row built so far.  But it FORGETS: whenever row i is not
    // merged, the row built so far is finished and the walk goes on from row i alone -- its state then is a function of i.  So the
    // row sequence is cut into ranges of equal interval count and every range is walked by a host thread that ASSUMES such a
    // fresh start at its first row; two walks that start afresh at the same row are the same walk from there on.  Afterwards the
    // ranges are stitched in order: the true walk (thread 0's; later the walk carried over a boundary) continues into the next
    // range until it starts afresh at a row where that range's own walk did too -- a few rows, fresh starts come every four rows
    // on average -- and from that row on the range's finished rows are the true ones.  Same rows, same order as one thread's
    // walk (rounds 4 - 5: 138 ms at n = 1e8, Wu = 50; the sparse rows at both ends of the key range are one unbroken chain of
    // small rows each, so cutting only at rows the rule can never merge left two threads with twice the work of the others).
    // Finished rows never hold more intervals than the rows they were made of (a union is kept only if it is shorter than
    // 0.8 x its parts), so a range's thread writes its finished rows into ONE buffer of `cnt` intervals from the position of its
    // first row on; what the stitching finishes goes into small side buffers.
    // (the buffer is kept between calls -- a process builds the five indexes of a series one after the other, and giving
    // 400 MB back to the system and faulting them in again cost 50 ms per call; a call that finds it taken allocates its own)
    struct kept_buffer {
        std::mutex mu;
        std::unique_ptr<ival[]> p;
        size_t cap = 0;
    };
    static kept_buffer *kept = new kept_buffer();   // (never destroyed: a call may be running at exit)
    std::unique_lock<std::mutex> kept_lock(kept->mu, std::try_to_lock);
    std::unique_ptr<ival[]> own;
    ival *stored;
    if (kept_lock.owns_lock()) {
        if (kept->cap < (size_t)cnt + 1) {
            kept->p.reset();
            kept->p.reset(new ival[(size_t)cnt + 1]);
            kept->cap = (size_t)cnt + 1;
        }
        stored = kept->p.get();
    } else {
        own.reset(new ival[(size_t)cnt + 1]);
        stored = own.get();
    }
    struct row_t {
        const ival *p;       // the finished row's intervals
        int64_t n;
        double key;
        int64_t first_row;   // the row of step 1 the walk started this row at
        int64_t line_len = -1;   // bytes of the row's line in the file and the offsets it covers, where the range's own thread had
        uint32_t offsets = 0;    // the time to work them out (below); -1: left to the encoding passes
        const uint8_t *line = nullptr;   // ... and the line itself, encoded ahead into the range's side buffer
    };
    auto row_line_facts = [](row_t &r) {
        r.line_len = 8 + encode_compact<false>(r.p, r.n, nullptr);
        uint32_t nofs = 0;   // (a Java int sum: wraps like it)
        for (int64_t q = 0; q < r.n; q++) nofs += (uint32_t)(r.p[q].last - r.p[q].first + 1);
        r.offsets = nofs;
    };
    struct chain {           // the walk's state: the row being built
        std::vector<ival> acc, merged;
        bool last_is_raw = true;   // it is still row g_last as step 1 left it (a view into the grouped arrays), not a union (`acc`)
        int64_t g_last = 0, first_row = 0;
    };
    auto raw = [&](int64_t g) { return raw_view{rf + goff[g], rl + goff[g], (int64_t)goff[g + 1] - goff[g]}; };
    auto chain_size = [&](const chain &c) { return c.last_is_raw ? raw(c.g_last).n : (int64_t)c.acc.size(); };
    auto chain_copy = [&](const chain &c, ival *dst) {
        if (c.last_is_raw) {
            const raw_view v = raw(c.g_last);
            for (int64_t i = 0; i < v.n; i++) dst[i] = v.at(i);
        } else {
            std::memcpy(dst, c.acc.data(), c.acc.size() * sizeof(ival));
        }
    };
    // row i meets the chain: merged into it -- or the chain's row is finished (finish(key)) and row i starts the next
    auto step = [&](chain &c, int64_t i, auto &&finish) {
        const raw_view cur = raw(i);
        bool is_merged = false;
        if ((double)cur.n < average * 1.2) {
            // the union, built in one pass (IndexNodeUtils.addInterval cuts every piece to <= 256 positions) and kept
            // only if it is short enough
            const int64_t last_n = chain_size(c);
            std::vector<ival> &merged = c.merged;
            merged.clear();
            auto emit = [&](int32_t a, int32_t b) {
                while (b - a >= MAXIMUM_DIFF) {
                    merged.push_back(ival{a, a + MAXIMUM_DIFF - 1});
                    a += MAXIMUM_DIFF;
                }
                merged.push_back(ival{a, b});
            };
            // (a run of untouched intervals: those of a union are cut already, those of a step-1 row span at most 255
            // positions -- addInterval would hand each of them on as it is)
            auto bulk_raw = [&](const raw_view &v, int64_t i0, int64_t j) {
                for (int64_t t = i0; t < j; t++) merged.push_back(v.at(t));
            };
            auto bulk_vec = [&](const vec_view &v, int64_t i0, int64_t j) { merged.insert(merged.end(), v.p + i0, v.p + j); };
            if (c.last_is_raw) merge_walk(raw(c.g_last), cur, emit, bulk_raw, bulk_raw);
            else merge_walk(vec_view{c.acc.data(), (int64_t)c.acc.size()}, cur, emit, bulk_vec, bulk_raw);
            if ((double)merged.size() < (double)(last_n + cur.n) * 0.8) {
                c.acc.swap(merged);
                c.last_is_raw = false;
                is_merged = true;
            }
        }
        if (!is_merged) {
            finish(gkey[i - 1]);
            c.last_is_raw = true;
            c.g_last = c.first_row = i;
        }
        return !is_merged;   // the walk started afresh at row i
    };
    struct walked {
        int64_t b0 = 0, b1 = 0;    // rows [b0, b1)
        std::vector<row_t> rows;   // finished, in order
        std::unique_ptr<uint8_t[]> lines;   // their lines, encoded while the end ranges were still being walked
        chain pending;             // the row being built when the range ended
    };
    // (threads: at most 16 -- the ranges' own, less what the two teams of the sparse ends add, see run_team below)
    const int hc = (int)std::max(1u, std::thread::hardware_concurrency());
    // (measured at n = 1e8 on 64 cores in 8 cache domains, Wu = 25, step 2: one thread per end range 99 ms; teams wherever the
    // scheduler put them: 2 threads 71 ms, 4 threads 65 ms; on CPUs that share a cache (cache_mates): 2 threads 54 ms, 3 threads
    // 45 ms, 4 threads 40 ms)
    const int team = g_team_size.load() > 0 ? (int)std::min<int64_t>(g_team_size.load(), 8) : (hc >= 16 ? 4 : (hc >= 4 ? 2 : 1));
    const int nw = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)hc, 16 - 2 * (team - 1), G / 64 + 1}));
    std::vector<walked> parts(1);
    for (int t = 1; t < nw; t++) {
        // ranges of about equal interval count, cut where the true walk certainly starts afresh: at the first row behind the
        // share point that is never merged (a row of at least 1.2 x the average); the stitching then joins at once.  The sparse
        // rows at both ends of the key range form chains without such a row, in which two walks started at different rows take
        // thousands of rows to meet (measured: 2 932 and 1 824 rows at n = 1e8, Wu = 25): a cut inside them buys nothing -- the
        // true walk redoes the range -- so those chains stay whole (86 of this step's 93 ms were there; they are walked as a
        // pipeline now: run_team).
        int64_t c = (int64_t)(std::lower_bound(goff, goff + G, (int32_t)(cnt * t / nw)) - goff);
        while (c < G && !((double)((int64_t)goff[c + 1] - goff[c]) >= average * 1.2)) c++;
        if (c > parts.back().b0 && c < G) {
            parts.back().b1 = c;
            parts.emplace_back();
            parts.back().b0 = c;
        }
    }
    parts.back().b1 = G;
    // Ranges of many rows (the sparse ends of the key range: a third of the rows, a tenth of the intervals, nine tenths of the
    // work) are walked by a team of threads as a pipeline (pipe_rec above); the others by one thread each.
    const int64_t TEAM_MIN_ROWS = g_team_min_rows.load(), TEAM_MIN_ROW_SIZE = g_team_min_row_size.load();
    std::atomic<int> pipe_error{0};
    auto run_team = [&](size_t t, int D) {
        walked &o = parts[t];
        const double w0 = now_ms();
        const int64_t b0 = o.b0, b1 = o.b1, nr = b1 - b0;
        const int NB = D + 1;   // buffers: row i writes buffer i mod NB, read by row i + 1 alone -- whose thread takes row i + NB next
        int64_t maxrow = 0;
        for (int64_t i = b0; i < b1; i++) maxrow = std::max<int64_t>(maxrow, (int64_t)goff[i + 1] - goff[i]);
        // a union the walk keeps holds fewer than max(largest row, 4.8 x average) intervals (it is kept only below 0.8 x (union
        // before + row), row < 1.2 x average); a union made on a wrong guess, up to D guesses deep, that + D + 1 rows
        const int64_t cap = std::max<int64_t>(maxrow, (int64_t)(average * 4.8) + 1) + (int64_t)(average * 1.2 + 1) * (D + 2) + 64;
        std::unique_ptr<std::atomic<uint64_t>[]> slab(new std::atomic<uint64_t>[(size_t)(cap * NB)]);
        std::unique_ptr<pipe_rec[]> rec(new pipe_rec[(size_t)nr]);
        for (int64_t q = 0; q < nr; q++) rec[(size_t)q].buf = slab.get() + ((b0 + q) % NB) * cap;
        ival *at = stored + goff[b0];   // (handed from task to task with the decisions, like o.rows)
        // the walk's row behind row i - 1 is finished under row i - 1's key
        auto finish_behind = [&](int64_t i) {
            const pipe_rec &pr = rec[(size_t)(i - 1 - b0)];
            const int64_t nn = pr.n;
            if (pr.decided.load(std::memory_order_relaxed) == 2) {
                const raw_view v = raw(i - 1);
                for (int64_t q = 0; q < nn; q++) at[q] = v.at(q);
            } else {
                for (int64_t q = 0; q < nn; q++) at[q] = pipe_unpack(pr.buf[q].load(std::memory_order_relaxed));
            }
            o.rows.push_back(row_t{at, nn, gkey[i - 1], pr.first_row});
            at += nn;
        };
        auto wait_decided = [&](const pipe_rec &pr) {
            int spins = 0;
            while (pr.decided.load(std::memory_order_acquire) == 0) {
                if (pipe_error.load(std::memory_order_relaxed)) return false;
                pipe_relax(spins);
            }
            return true;
        };
        rec[0].n = (int64_t)goff[b0 + 1] - goff[b0];
        rec[0].first_row = b0;
        rec[0].decided.store(2, std::memory_order_release);
        auto member = [&](int d) {
            try {
                for (int64_t i = b0 + 1 + d; i < b1; i += D) {
                    pipe_rec &me = rec[(size_t)(i - b0)];
                    const pipe_rec &pr = rec[(size_t)(i - 1 - b0)];
                    const raw_view cur = raw(i);
                    bool kept = false;
                    int64_t mn = 0;
                    if ((double)cur.n < average * 1.2) {
                        uint64_t attempt = 0;
                        bool guess = true;   // merge into the predecessor's union while it is still being made
                        for (;;) {
                            if (pipe_error.load(std::memory_order_relaxed)) return;
                            if (!guess && !wait_decided(pr)) return;
                            attempt++;
                            me.prog.store(attempt << 40, std::memory_order_release);   // (whoever reads my union starts again)
                            pipe_sink out{&me, cap, 0, attempt};
                            int rc;
                            if (pr.decided.load(std::memory_order_acquire) == 2) {
                                pipe_src_raw src{raw(i - 1)};
                                rc = merge_from(src, cur, out);
                            } else {
                                pipe_src_stream src{&pr, pr.prog.load(std::memory_order_acquire) >> 40};
                                rc = merge_from(src, cur, out);
                            }
                            if (rc == 0) {
                                mn = out.o;
                                me.prog.store((attempt << 40) | (uint64_t)mn, std::memory_order_release);
                                break;
                            }
                            if (rc == -2) {
                                if (!guess) {   // (cannot happen: see `cap`)
                                    pipe_error.store(1);
                                    return;
                                }
                                guess = false;
                            }
                        }
                        if (!wait_decided(pr)) return;   // (it is: the source came to its end)
                        kept = (double)mn < (double)(pr.n + cur.n) * 0.8;
                    } else if (!wait_decided(pr)) {
                        return;
                    }
                    if (kept) {
                        me.n = mn;
                        me.first_row = pr.first_row;
                        me.decided.store(1, std::memory_order_release);
                    } else {
                        finish_behind(i);
                        me.n = cur.n;
                        me.first_row = i;
                        me.decided.store(2, std::memory_order_release);
                    }
                }
            } catch (...) {
                pipe_error.store(2);
            }
        };
        {
            // the team on CPUs that share a cache (threads inherit the mask of the thread that starts them); this thread's own
            // mask -- it may be the caller's -- is put back afterwards
            cpu_set_t mine, mates;
            const bool pinned = pthread_getaffinity_np(pthread_self(), sizeof mine, &mine) == 0 && cache_mates(&mates) &&
                                pthread_setaffinity_np(pthread_self(), sizeof mates, &mates) == 0;
            std::vector<std::thread> th;
            try {
                for (int d = 1; d < D; d++) th.emplace_back(member, d);
            } catch (...) {
                pipe_error.store(2);
            }
            member(0);
            for (auto &x : th) x.join();
            if (pinned) (void)pthread_setaffinity_np(pthread_self(), sizeof mine, &mine);
        }
        if (pipe_error.load()) return;
        // the walk's state at the end of the range
        const pipe_rec &lr = rec[(size_t)(nr - 1)];
        chain &c = o.pending;
        c.first_row = lr.first_row;
        if (lr.decided.load() == 2) {
            c.last_is_raw = true;
            c.g_last = b1 - 1;
        } else {
            c.last_is_raw = false;
            c.acc.resize((size_t)lr.n);
            for (int64_t q = 0; q < lr.n; q++) c.acc[(size_t)q] = pipe_unpack(lr.buf[q].load(std::memory_order_relaxed));
        }
        if (getenv("KVM_TRACE"))
            fprintf(stderr, "[kvm] index walk range %d rows %lld..%lld intervals %lld: %.1f ms (a team of %d)\n", (int)t, (long long)b0, (long long)b1,
                    (long long)(goff[b1] - goff[b0]), now_ms() - w0, D);
    };
    {
        auto run = [&](size_t t) {
            walked &o = parts[t];
            const double w0 = now_ms();
            chain &c = o.pending;
            c.g_last = c.first_row = o.b0;
            ival *at = stored + goff[o.b0];
            for (int64_t i = o.b0 + 1; i < o.b1; i++)
                step(c, i, [&](double key) {
                    const int64_t nn = chain_size(c);
                    chain_copy(c, at);
                    o.rows.push_back(row_t{at, nn, key, c.first_row});
                    at += nn;
                });
            const double w1 = now_ms();
            // (the ranges between the sparse ends are walked in a few milliseconds; the teams at the ends take ten times as long:
            // this thread measures its own rows' lines meanwhile -- nine tenths of the intervals)
            int64_t bytes = 0;
            for (row_t &r : o.rows) {
                row_line_facts(r);
                bytes += r.line_len;
            }
            // ... and encodes them: where a line goes in the file depends on the lines of the lowest keys -- the far end's, finished
            // last -- so into a buffer of its own; the final pass then copies (a tenth of what encoding costs)
            o.lines.reset(new uint8_t[(size_t)bytes + 1]);
            uint8_t *dst = o.lines.get();
            for (row_t &r : o.rows) {
                put_be_double(dst, r.key);
                (void)encode_compact<true>(r.p, r.n, dst + 8);
                r.line = dst;
                dst += r.line_len;
            }
            if (getenv("KVM_TRACE"))
                fprintf(stderr, "[kvm] index walk thread %d rows %lld..%lld intervals %lld: %.1f ms + %.1f ms encoding its lines\n", (int)t, (long long)o.b0,
                        (long long)o.b1, (long long)(goff[o.b1] - goff[o.b0]), w1 - w0, now_ms() - w1);
        };
        auto walk = [&](size_t t) {
            try {   // (a worker thread: nothing may leave it)
                const int64_t nrows = parts[t].b1 - parts[t].b0, nint = (int64_t)goff[parts[t].b1] - goff[parts[t].b0];
                if (team > 1 && nrows >= TEAM_MIN_ROWS && nint >= TEAM_MIN_ROW_SIZE * nrows) run_team(t, team);
                else run(t);
            } catch (...) {
                pipe_error.store(2);
            }
        };
        std::vector<std::thread> th;
        for (size_t t = 1; t < parts.size(); t++) th.emplace_back(walk, t);
        walk(0);
        for (auto &x : th) x.join();
    }
    if (pipe_error.load()) {
        kvm_index_rows_free(rows);
        delete f;
        kvm_set_error(ctx, "kvm_index_build: step 2 failed (out of memory, or an internal error of the pipelined walk)");
        return pipe_error.load() == 2 ? KVM_ERR_NO_MEMORY : KVM_ERR_DEVICE;
    }
    // ---- stitching (one thread; a few rows per boundary)
    std::vector<row_t> rows_done;
    std::deque<std::vector<ival>> side;   // rows finished while carrying the true walk over a boundary (stable addresses)
    chain *truth = &parts[0].pending;      // the true walk's state at the end of the ranges stitched so far
    rows_done = parts[0].rows;
    auto finish_side = [&](chain &c, double key) {
        side.emplace_back((size_t)chain_size(c));
        chain_copy(c, side.back().data());
        rows_done.push_back(row_t{side.back().data(), (int64_t)side.back().size(), key, c.first_row});
    };
    for (size_t t = 1; t < parts.size(); t++) {
        walked &o = parts[t];
        bool joined = false;
        const double s0 = now_ms();
        int64_t i_join = o.b1;
        for (int64_t i = o.b0; i < o.b1 && !joined; i++) {
            i_join = i;
            const bool fresh = step(*truth, i, [&](double key) { finish_side(*truth, key); });
            if (!fresh) continue;
            // the true walk starts afresh at row i: did this range's own walk as well?  (its finished rows are ordered by first_row)
            auto it = std::lower_bound(o.rows.begin(), o.rows.end(), i, [](const row_t &r, int64_t v) { return r.first_row < v; });
            if (it != o.rows.end() && it->first_row == i) {
                rows_done.insert(rows_done.end(), it, o.rows.end());
                truth = &o.pending;
                joined = true;
            } else if (o.pending.first_row == i) {
                truth = &o.pending;
                joined = true;
            }
        }
        // (not joined: the true walk crossed the whole range without a common fresh start -- it simply goes on into the next)
        if (getenv("KVM_TRACE"))
            fprintf(stderr, "[kvm] index stitch into range %d (rows %lld..%lld, %zu finished rows): %s after %lld rows, %.1f ms\n", (int)t, (long long)o.b0,
                    (long long)o.b1, o.rows.size(), joined ? "joined" : "NOT joined", (long long)(i_join - o.b0 + 1), now_ms() - s0);
    }
    finish_side(*truth, gkey[G - 1]);   // the end of the walk: the row being built is finished under the last key
    kvm_index_rows_free(rows);
    const int64_t R = (int64_t)rows_done.size();
    int64_t total_iv = 0;
    for (const row_t &r : rows_done) total_iv += r.n;
    f->rows_after = R;
    f->intervals_after = total_iv;
    const double t2 = now_ms();
    f->step2_ms = t2 - t1;

    // ---- the file image: TreeMap order = ascending key = the finished rows backwards: line k is finished row R - 1 - k.  The
    // lines are independent of each other.  Two passes by the same host threads over contiguous ranges of lines balanced by
    // intervals: the length of every line (the encoder without its stores), then -- the image allocated once, at its final
    // size, every offset known -- the lines encoded in place.
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 16, R / 64 + 1}));
    std::vector<int64_t> cut((size_t)nt + 1, R);
    cut[0] = 0;
    {
        // (by the work that is left: a line encoded ahead is a copy -- a sixteenth of an interval's cost per interval, say; the
        // lines of the end ranges, measured and encoded here, are what the threads have to share)
        auto cost = [](const row_t &r) { return r.line ? r.n / 16 + 1 : r.n + 1; };
        int64_t total_cost = 0, seen = 0;
        for (const row_t &r : rows_done) total_cost += cost(r);
        int t = 1;
        for (int64_t k = 0; k < R && t < nt; k++) {
            seen += cost(rows_done[(size_t)(R - 1 - k)]);
            if (seen >= total_cost * t / nt) cut[(size_t)t++] = k + 1;
        }
    }
    auto on_threads = [&](auto &&work) {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    };
    std::vector<int64_t> line_len((size_t)R);
    on_threads([&](int t) {
        for (int64_t k = cut[(size_t)t]; k < cut[(size_t)t + 1]; k++) {
            row_t &r = rows_done[(size_t)(R - 1 - k)];
            if (r.line_len < 0) row_line_facts(r);
            line_len[(size_t)k] = r.line_len;
        }
    });
    std::vector<int64_t> line_at((size_t)R + 1, 0);
    for (int64_t k = 0; k < R; k++) line_at[(size_t)k + 1] = line_at[(size_t)k] + line_len[(size_t)k];
    const int64_t lines_bytes = line_at[(size_t)R];
    const int64_t image_bytes = lines_bytes + 16 * R + 4 * (R + 2);   // lines, statistic triples, the offset table (R lines + 2 marks)
    if (image_bytes > 0x7fffffffLL) {
        delete f;
        kvm_set_error(ctx, "kvm_index_build: the index file would reach 2 GiB (its offsets are Java ints)");
        return KVM_ERR_UNSUPPORTED;
    }
    f->take_image((size_t)image_bytes);
    uint8_t *img = f->image.get();
    on_threads([&](int t) {
        for (int64_t k = cut[(size_t)t]; k < cut[(size_t)t + 1]; k++) {
            const row_t &r = rows_done[(size_t)(R - 1 - k)];
            uint8_t *dst = img + line_at[(size_t)k];
            if (r.line) {
                std::memcpy(dst, r.line, (size_t)r.line_len);
                continue;
            }
            put_be_double(dst, r.key);
            (void)encode_compact<true>(r.p, r.n, dst + 8);
        }
    });
    // the statistic triples {key, running #intervals, running #offsets} in line order (Java int sums: they wrap like them);
    // the per-row offset counts were taken with the line lengths, the running sums here
    uint32_t cum_iv = 0, cum_off = 0;
    for (int64_t k = 0; k < R; k++) {
        const row_t &r = rows_done[(size_t)(R - 1 - k)];
        cum_iv += (uint32_t)r.n;
        cum_off += r.offsets;
        uint8_t *at = img + lines_bytes + 16 * k;
        put_be_double(at, r.key);
        put_be32(at + 8, (int32_t)cum_iv);
        put_be32(at + 12, (int32_t)cum_off);
    }
    // the table of line offsets; its last two entries: where the statistic triples begin, and its own position
    {
        uint8_t *tab = img + lines_bytes + 16 * R;
        for (int64_t k = 0; k < R; k++) put_be32(tab + 4 * k, (int32_t)line_at[(size_t)k]);
        put_be32(tab + 4 * R, (int32_t)lines_bytes);
        put_be32(tab + 4 * (R + 1), (int32_t)(lines_bytes + 16 * R));
    }
    f->encode_ms = now_ms() - t2;
    if (getenv("KVM_TRACE")) fprintf(stderr, "[kvm] index build: step1 %.1f step2 %.1f encode %.1f, buffers freed after %.1f ms of the call\n", f->step1_ms, f->step2_ms, f->encode_ms, now_ms() - t0);
    *out = f;
    return KVM_OK;
    });
}

extern "C" int64_t kvm_index_file_size(const kvm_index_file *f) { return f ? (int64_t)f->image_size : 0; }
extern "C" const uint8_t *kvm_index_file_bytes(const kvm_index_file *f) { return f ? f->image.get() : nullptr; }
extern "C" int64_t kvm_index_file_info(const kvm_index_file *f, int32_t what) {
    if (!f) return 0;
    switch (what) {
        case KVM_INDEX_ROWS_BEFORE: return f->rows_before;
        case KVM_INDEX_ROWS_AFTER: return f->rows_after;
        case KVM_INDEX_INTERVALS_BEFORE: return f->intervals_before;
        case KVM_INDEX_INTERVALS_AFTER: return f->intervals_after;
        case KVM_INDEX_STEP1_US: return (int64_t)(f->step1_ms * 1e3);
        case KVM_INDEX_STEP2_US: return (int64_t)(f->step2_ms * 1e3);
        case KVM_INDEX_ENCODE_US: return (int64_t)(f->encode_ms * 1e3);
    }
    return 0;
}
extern "C" int kvm_index_file_write(const kvm_index_file *f, const char *path) {
    return kvm_guard<int>(nullptr, [&]() -> int {
    if (!f || !path) return KVM_ERR_ILLEGAL_ARGUMENT;
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return KVM_ERR_IO;
    const size_t n = f->image_size;
    const bool ok = std::fwrite(f->image.get(), 1, n, fp) == n;
    return (std::fclose(fp) == 0 && ok) ? KVM_OK : KVM_ERR_IO;
    });
}
extern "C" void kvm_index_file_free(kvm_index_file *f) { delete f; }

// ---- sortAndMergeIntervals (K/QueryEngine.java:664-693): the step between phase 1 and phase 2 -------------------------
extern "C" int64_t kvm_sort_and_merge_intervals(const int32_t *left, const int32_t *right, int64_t k, int32_t *out_left,
                                                int32_t *out_right) {
    return kvm_guard<int64_t>(nullptr, [&]() -> int64_t {
    if (k < 0 || (k > 0 && (!left || !right || !out_left || !out_right))) return KVM_ERR_ILLEGAL_ARGUMENT;
    std::vector<ival> v((size_t)k);
    for (int64_t i = 0; i < k; i++) v[(size_t)i] = ival{left[i], right[i]};
    if (k > 1) {
        // intervals.sort(Comparator.comparingInt(Interval::getLeft)): List.sort is stable
        std::stable_sort(v.begin(), v.end(), [](const ival &a, const ival &b) { return a.first < b.first; });
        int64_t o = 0;
        int32_t start = v[0].first, end = v[0].last;
        for (int64_t i = 1; i < k; i++) {
            const ival c = v[(size_t)i];
            if ((int64_t)c.first - 1 <= end) {
                end = std::max(c.last, end);
            } else {
                v[(size_t)o++] = ival{start, end};
                start = c.first;
                end = c.last;
            }
        }
        v[(size_t)o++] = ival{start, end};
        k = o;
    }
    for (int64_t i = 0; i < k; i++) {
        out_left[i] = v[(size_t)i].first;
        out_right[i] = v[(size_t)i].last;
    }
    return k;
    });
}
