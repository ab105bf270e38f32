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
// The walk is sequential by nature (each decision depends on the row merged so far): one pass per attempted union, read
// straight from the grouped step-1 arrays; the lines of the file are encoded by a few host threads.
#include "kvm_guard.h"
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kvmatch.h"

void kvm_set_error(kvm_ctx *ctx, const std::string &msg);  // kvm_api.hip

struct kvm_index_file {
    std::vector<uint8_t> image;
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

// IndexNodeUtils.mergeIndexNode; Emit(first, last) receives every interval handed to addInterval
template <class N1, class N2, class Emit>
inline void merge_walk(const N1 &n1, const N2 &n2, Emit emit) {
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
        } else if (l2.last + 1 < l1.first) {
            emit(l2.first, l2.last);
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

inline void put_be32(uint8_t *p, int32_t v) {
    const uint32_t u = (uint32_t)v;
    p[0] = (uint8_t)(u >> 24); p[1] = (uint8_t)(u >> 16); p[2] = (uint8_t)(u >> 8); p[3] = (uint8_t)u;
}
inline void put_be_double(uint8_t *p, double v) {
    uint64_t u;
    std::memcpy(&u, &v, 8);
    for (int b = 0; b < 8; b++) p[b] = (uint8_t)(u >> (56 - 8 * b));
}

// IndexNode.toBytesCompact; appends to `out`
void encode_compact(const ival *pos, int64_t size, std::vector<uint8_t> &out) {
    const size_t base = out.size();
    out.resize(base + (size_t)size * 8);
    uint8_t *result = out.data() + base;
    int64_t index = 0, length = 0;
    int count = 0;
    bool packing = false;
    while (index < size) {
        if (!packing) {
            put_be32(result + length, pos[index].first);
            length += 4 + 1;
            result[length++] = (uint8_t)(pos[index].last - pos[index].first - 128);
            packing = true;
            count = 1;
        } else {
            const int32_t gap = pos[index].first - pos[index - 1].last;
            if (gap < MAXIMUM_DIFF && (count - 1) / 2 + 2 < MAXIMUM_DIFF) {
                result[length++] = (uint8_t)(gap - 128);
                result[length++] = (uint8_t)(pos[index].last - pos[index].first - 128);
                count += 2;
            } else {
                result[length - count - 1] = (uint8_t)((count - 1) / 2 - 128);
                packing = false;
                continue;
            }
        }
        index++;
    }
    if (packing) result[length - count - 1] = (uint8_t)((count - 1) / 2 - 128);
    out.resize(base + (size_t)length);
}

double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

}  // namespace

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
    int rc = kvm_index_rows_build_grouped(ctx, w, &rows);
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
    std::vector<ival> stored;                 // intervals of the finished rows, in the order they are finished
    std::vector<int64_t> stored_off{0};       // row r = stored[stored_off[r], stored_off[r+1])
    std::vector<double> stored_key;
    stored.reserve((size_t)cnt);
    // `last` is either row g_last as step 1 left it (a view into the grouped arrays) or the union built so far (`acc`)
    std::vector<ival> acc, merged;
    bool last_is_raw = true;
    int64_t g_last = 0;
    auto raw = [&](int64_t g) { return raw_view{rf + goff[g], rl + goff[g], (int64_t)goff[g + 1] - goff[g]}; };
    auto store = [&](double key) {
        if (last_is_raw) {
            const raw_view v = raw(g_last);
            for (int64_t i = 0; i < v.n; i++) stored.push_back(v.at(i));
        } else {
            stored.insert(stored.end(), acc.begin(), acc.end());
        }
        stored_off.push_back((int64_t)stored.size());
        stored_key.push_back(key);
    };
    for (int64_t i = 1; i < G; i++) {
        const raw_view cur = raw(i);
        bool is_merged = false;
        if ((double)cur.n < average * 1.2) {
            // the union, built in one pass (IndexNodeUtils.addInterval cuts every piece to <= 256 positions) and kept
            // only if it is short enough
            const int64_t last_n = last_is_raw ? raw(g_last).n : (int64_t)acc.size();
            merged.clear();
            auto emit = [&](int32_t a, int32_t b) {
                while (b - a >= MAXIMUM_DIFF) {
                    merged.push_back(ival{a, a + MAXIMUM_DIFF - 1});
                    a += MAXIMUM_DIFF;
                }
                merged.push_back(ival{a, b});
            };
            if (last_is_raw) merge_walk(raw(g_last), cur, emit);
            else merge_walk(vec_view{acc.data(), (int64_t)acc.size()}, cur, emit);
            if ((double)merged.size() < (double)(last_n + cur.n) * 0.8) {
                acc.swap(merged);
                last_is_raw = false;
                is_merged = true;
            }
        }
        if (!is_merged) {
            store(gkey[i - 1]);
            last_is_raw = true;
            g_last = i;
        }
    }
    store(gkey[G - 1]);
    kvm_index_rows_free(rows);
    const int64_t R = (int64_t)stored_key.size();
    f->rows_after = R;
    f->intervals_after = (int64_t)stored.size();
    const double t2 = now_ms();
    f->step2_ms = t2 - t1;

    // ---- the file image: TreeMap order = ascending key = the finished rows backwards.  The lines are independent of
    // each other: encoded by a few host threads into per-thread buffers (contiguous ranges of lines), then laid end to end.
    std::vector<uint8_t> &img = f->image;
    std::vector<int32_t> offsets;
    offsets.reserve((size_t)R + 2);
    bool too_long = false;
    {
        const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 8, R / 64 + 1}));
        std::vector<std::vector<uint8_t>> part((size_t)nt);
        std::vector<std::vector<int64_t>> line_len((size_t)nt);
        // balance by intervals: line k (ascending key) is stored row R-1-k
        std::vector<int64_t> cut((size_t)nt + 1, R);
        cut[0] = 0;
        {
            const int64_t total = (int64_t)stored.size();
            int64_t seen = 0;
            int t = 1;
            for (int64_t k = 0; k < R && t < nt; k++) {
                seen += stored_off[(size_t)(R - 1 - k) + 1] - stored_off[(size_t)(R - 1 - k)];
                if (seen >= total * t / nt) cut[(size_t)t++] = k + 1;
            }
        }
        auto work = [&](int t) {
            std::vector<uint8_t> &out = part[(size_t)t];
            for (int64_t k = cut[(size_t)t]; k < cut[(size_t)t + 1]; k++) {
                const int64_t r = R - 1 - k;
                const size_t at = out.size();
                out.resize(at + 8);
                put_be_double(out.data() + at, stored_key[(size_t)r]);
                encode_compact(stored.data() + stored_off[(size_t)r], stored_off[(size_t)r + 1] - stored_off[(size_t)r], out);
                line_len[(size_t)t].push_back((int64_t)(out.size() - at));
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        size_t total_bytes = 0;
        for (auto &v : part) total_bytes += v.size();
        if (total_bytes > 0x7fffffffULL) too_long = true;
        if (!too_long) {
            img.reserve(total_bytes + (size_t)R * 20 + 64);
            for (int t = 0; t < nt; t++) {
                size_t at = img.size();
                for (int64_t len : line_len[(size_t)t]) {
                    offsets.push_back((int32_t)at);
                    at += (size_t)len;
                }
                img.insert(img.end(), part[(size_t)t].begin(), part[(size_t)t].end());
            }
        }
    }
    auto mark = [&]() {
        if (img.size() > 0x7fffffffULL) too_long = true;
        offsets.push_back((int32_t)img.size());
    };
    mark();
    uint32_t cum_iv = 0, cum_off = 0;  // Java int sums (wrap like them)
    for (int64_t r = R - 1; r >= 0 && !too_long; r--) {
        uint32_t nofs = 0;
        for (int64_t t = stored_off[(size_t)r]; t < stored_off[(size_t)r + 1]; t++)
            nofs += (uint32_t)(stored[(size_t)t].last - stored[(size_t)t].first + 1);
        cum_iv += (uint32_t)(stored_off[(size_t)r + 1] - stored_off[(size_t)r]);
        cum_off += nofs;
        const size_t at = img.size();
        img.resize(at + 16);
        put_be_double(img.data() + at, stored_key[(size_t)r]);
        put_be32(img.data() + at + 8, (int32_t)cum_iv);
        put_be32(img.data() + at + 12, (int32_t)cum_off);
    }
    mark();
    if (too_long || img.size() + offsets.size() * 4 > 0x7fffffffULL) {
        delete f;
        kvm_set_error(ctx, "kvm_index_build: the index file would reach 2 GiB (its offsets are Java ints)");
        return KVM_ERR_UNSUPPORTED;
    }
    {
        const size_t at = img.size();
        img.resize(at + offsets.size() * 4);
        for (size_t t = 0; t < offsets.size(); t++) put_be32(img.data() + at + t * 4, offsets[t]);
    }
    f->encode_ms = now_ms() - t2;
    *out = f;
    return KVM_OK;
    });
}

extern "C" int64_t kvm_index_file_size(const kvm_index_file *f) { return f ? (int64_t)f->image.size() : 0; }
extern "C" const uint8_t *kvm_index_file_bytes(const kvm_index_file *f) { return f ? f->image.data() : nullptr; }
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
    const size_t n = f->image.size();
    const bool ok = std::fwrite(f->image.data(), 1, n, fp) == n;
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
