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
// The walk is sequential by nature (each decision depends on the row merged so far); the union is first COUNTED and only
// materialised when it is accepted.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
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

// IndexNodeUtils.mergeIndexNode; Emit(first, last) receives every interval handed to addInterval
template <class Emit>
inline void merge_walk(const ival *n1, int64_t s1, const ival *n2, int64_t s2, Emit emit) {
    int64_t i1 = 0, i2 = 0;
    bool h1 = false, h2 = false;
    ival l1{0, 0}, l2{0, 0};
    while (i1 < s1 && i2 < s2) {
        if (!h1) { l1 = n1[i1]; h1 = true; }
        if (!h2) { l2 = n2[i2]; h2 = true; }
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
        if (!h1) l1 = n1[i];
        emit(l1.first, l1.last);
        h1 = false;
    }
    for (int64_t i = i2; i < s2; i++) {
        if (!h2) l2 = n2[i];
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
    std::vector<ival> last, cur, merged;
    auto load = [&](int64_t g, std::vector<ival> &v) {
        const int64_t b = goff[g], e = goff[g + 1];
        v.resize((size_t)(e - b));
        for (int64_t i = b; i < e; i++) v[(size_t)(i - b)] = ival{rf[i], rl[i]};
    };
    auto store = [&](double key) {
        stored.insert(stored.end(), last.begin(), last.end());
        stored_off.push_back((int64_t)stored.size());
        stored_key.push_back(key);
    };
    load(0, last);
    for (int64_t i = 1; i < G; i++) {
        load(i, cur);
        bool is_merged = false;
        if ((double)cur.size() < average * 1.2) {
            int64_t msize = 0;  // merged.getPositions().size(), counted: addInterval cuts [f, l] into (l - f) / 256 + 1 pieces
            merge_walk(last.data(), (int64_t)last.size(), cur.data(), (int64_t)cur.size(),
                       [&](int32_t a, int32_t b) { msize += (b - a) / MAXIMUM_DIFF + 1; });
            if ((double)msize < (double)(last.size() + cur.size()) * 0.8) {
                merged.clear();
                merged.reserve((size_t)msize);
                merge_walk(last.data(), (int64_t)last.size(), cur.data(), (int64_t)cur.size(), [&](int32_t a, int32_t b) {
                    while (b - a >= MAXIMUM_DIFF) {  // IndexNodeUtils.addInterval
                        merged.push_back(ival{a, a + MAXIMUM_DIFF - 1});
                        a += MAXIMUM_DIFF;
                    }
                    merged.push_back(ival{a, b});
                });
                last.swap(merged);
                is_merged = true;
            }
        }
        if (!is_merged) {
            store(gkey[i - 1]);
            last.swap(cur);
        }
    }
    store(gkey[G - 1]);
    kvm_index_rows_free(rows);
    const int64_t R = (int64_t)stored_key.size();
    f->rows_after = R;
    f->intervals_after = (int64_t)stored.size();
    const double t2 = now_ms();
    f->step2_ms = t2 - t1;

    // ---- the file image: TreeMap order = ascending key = the finished rows backwards
    std::vector<uint8_t> &img = f->image;
    img.reserve(stored.size() * 2 + (size_t)R * 32 + 64);
    std::vector<int32_t> offsets;
    offsets.reserve((size_t)R + 2);
    bool too_long = false;
    auto mark = [&]() {
        if (img.size() > 0x7fffffffULL) too_long = true;
        offsets.push_back((int32_t)img.size());
    };
    for (int64_t r = R - 1; r >= 0 && !too_long; r--) {
        mark();
        const size_t at = img.size();
        img.resize(at + 8);
        put_be_double(img.data() + at, stored_key[(size_t)r]);
        encode_compact(stored.data() + stored_off[(size_t)r], stored_off[(size_t)r + 1] - stored_off[(size_t)r], img);
    }
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
    if (!f || !path) return KVM_ERR_ILLEGAL_ARGUMENT;
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return KVM_ERR_IO;
    const size_t n = f->image.size();
    const bool ok = std::fwrite(f->image.data(), 1, n, fp) == n;
    return (std::fclose(fp) == 0 && ok) ? KVM_OK : KVM_ERR_IO;
}
extern "C" void kvm_index_file_free(kvm_index_file *f) { delete f; }

// ---- sortAndMergeIntervals (K/QueryEngine.java:664-693): the step between phase 1 and phase 2 -------------------------
extern "C" int64_t kvm_sort_and_merge_intervals(const int32_t *left, const int32_t *right, int64_t k, int32_t *out_left,
                                                int32_t *out_right) {
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
}
