// kvm_datagen.cpp -- seeded synthetic inputs (host side; used by tests and bench.py, not by the query path).
//
// kvm_gen_series restates the DISTRIBUTION of the reference's generator (K/DataGenerator.java:80-119 with
// K/data/RandomWalkGenerator.java:40-50, K/data/GaussianGenerator.java:53-84, K/data/SineGenerator.java:45-56,
// K/utils/RandomUtils.java:34-47).  The reference draws from unseeded ThreadLocalRandom / Math.random, so no
// run of it is reproducible; here one 64-bit seed fixes the whole series (xoshiro256** streams).
// kvm_gen_intervals produces phase-1-shaped candidate lists: sorted, disjoint, non-adjacent
// intervals with geometric lengths.
#include "kvm_guard.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/kvmatch.h"

namespace {

struct rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    explicit rng(uint64_t seed) {
        for (auto &v : s) v = splitmix(seed);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
    // RandomUtils.random(double min, double max) = nextDouble(min, max + 0.00001)
    double uniform(double lo, double hi) { return lo + unit() * ((hi + 0.00001) - lo); }
    // RandomUtils.random(int min, int max) = nextInt(min, max + 1)
    int64_t uniform_int(int64_t lo, int64_t hi) { return lo + (int64_t)(next() % (uint64_t)(hi - lo + 1)); }
};

void gen_random_walk(rng &g, double *out, int64_t len) {  // RandomWalkGenerator(-5, 5, 0, 1)
    out[0] = g.uniform(-5, 5);
    for (int64_t i = 1; i < len; i++) {
        double sign = g.unit() < 0.5 ? -1 : 1;
        double step = g.uniform(0, 1);
        out[i] = out[i - 1] + sign * step;
    }
}

void gen_gaussian(rng &g, double *out, int64_t len) {  // GaussianGenerator(-5, 5, 0, 2), Marsaglia polar
    double mean = g.uniform(-5, 5), sd = g.uniform(0, 2);
    double V1 = 0, V2 = 0, S = 0;
    int phase = 0;
    for (int64_t i = 0; i < len; i++) {
        double X;
        if (phase == 0) {
            do {
                V1 = 2.0 * g.unit() - 1.0;
                V2 = 2.0 * g.unit() - 1.0;
                S = V1 * V1 + V2 * V2;
            } while (S >= 1.0 || S == 0.0);
            X = V1 * std::sqrt(-2.0 * std::log(S) / S);
        } else {
            X = V2 * std::sqrt(-2.0 * std::log(S) / S);
        }
        phase = 1 - phase;
        out[i] = mean + sd * X;
    }
}

void gen_sine(rng &g, double *out, int64_t len) {  // SineGenerator(2, 10, 2, 10, -5, 5)
    double frequency = g.uniform(2, 10), amplitude = g.uniform(2, 10), mean = g.uniform(-5, 5);
    double phase = g.uniform(0, 2 * M_PI);
    for (int64_t i = 0; i < len; i++) {
        out[i] = mean + amplitude * std::sin(2 * i * (M_PI / len) * frequency + phase) +
                 g.uniform(amplitude * 0.05 * -1, amplitude * 0.05);
    }
}

}  // namespace

extern "C" int kvm_gen_series(uint64_t seed, int64_t n, double *out) {
    return kvm_guard<int>(nullptr, [&]() -> int {
    if (n < 1 || !out) return KVM_ERR_ILLEGAL_ARGUMENT;
    rng g(seed);
    int64_t left = n, pos = 0;
    while (left > 0) {
        // generateSegment(Math.min(left, dataLength / 100)), K/DataGenerator.java:80-86,106
        int64_t maxLength = std::min<int64_t>(left, n / 100);
        if (maxLength < 1) maxLength = left;  // n < 100: the reference would loop forever on 0; take the rest
        int t = (int)g.uniform_int(0, 2);
        int64_t l = g.uniform_int(std::min<int64_t>(1000, maxLength), maxLength);
        switch (t) {
            case 0: gen_random_walk(g, out + pos, l); break;
            case 1: gen_gaussian(g, out + pos, l); break;
            default: gen_sine(g, out + pos, l); break;
        }
        pos += l;
        left -= l;
    }
    return KVM_OK;
    });
}

extern "C" int64_t kvm_gen_intervals(uint64_t seed, int64_t n, int32_t m, int64_t target_candidates,
                                     double mean_len, const int32_t *must_cover, int32_t n_must, int32_t *left,
                                     int32_t *right, int64_t cap) {
    return kvm_guard<int64_t>(nullptr, [&]() -> int64_t {
    if (n < m || m < 1 || target_candidates < 0 || mean_len < 1 || !left || !right) return KVM_ERR_ILLEGAL_ARGUMENT;
    rng g(seed ^ 0x6b766d5f69766c73ULL);
    const int64_t last = n - m + 1;  // largest valid window start
    // Draw interval lengths (geometric, mean mean_len) until the target is met, then spread them over
    // [1, last] with random gaps >= 2 (non-adjacent), sorted by construction.
    std::vector<int64_t> lens;
    int64_t total = 0;
    const double p = 1.0 / mean_len;
    while (total < target_candidates) {
        double u = g.unit();
        int64_t len = 1 + (int64_t)std::floor(std::log(1.0 - u) / std::log(1.0 - (p < 1.0 ? p : 0.999999)));
        if (mean_len <= 1.0) len = 1;
        if (len > target_candidates - total) len = target_candidates - total;
        if (len < 1) len = 1;
        lens.push_back(len);
        total += len;
    }
    const int64_t k = (int64_t)lens.size();
    const int64_t need = total + 2 * k;  // each interval followed by a gap of at least 2 (1 would make them adjacent)
    if (need > last) return KVM_ERR_ILLEGAL_ARGUMENT;
    // distribute the slack `last - need` into k+1 gaps via sorted uniform cuts
    std::vector<int64_t> cuts((size_t)k);
    const int64_t slack = last - need;
    for (auto &c : cuts) c = slack > 0 ? (int64_t)(g.next() % (uint64_t)(slack + 1)) : 0;
    std::sort(cuts.begin(), cuts.end());
    struct iv { int64_t l, r; };
    std::vector<iv> ivs((size_t)k);
    int64_t used = 0;
    for (int64_t i = 0; i < k; i++) {
        int64_t start = 1 + cuts[(size_t)i] + used;
        ivs[(size_t)i] = {start, start + lens[(size_t)i] - 1};
        used += lens[(size_t)i] + 2;
    }
    // make sure the mandatory offsets (query position, planted copies) are candidates
    for (int32_t t = 0; t < n_must; t++) {
        int64_t o = must_cover[t];
        if (o < 1 || o > last) continue;
        bool covered = false;
        for (const auto &v : ivs)
            if (o >= v.l && o <= v.r) { covered = true; break; }
        if (!covered) ivs.push_back({o, o});
    }
    std::sort(ivs.begin(), ivs.end(), [](const iv &a, const iv &b) { return a.l < b.l; });
    // sortAndMergeIntervals semantics (K/QueryEngine.java:664-693): merge when current.left - 1 <= end
    std::vector<iv> merged;
    for (const auto &v : ivs) {
        if (!merged.empty() && v.l - 1 <= merged.back().r)
            merged.back().r = std::max(merged.back().r, v.r);
        else
            merged.push_back(v);
    }
    if ((int64_t)merged.size() > cap) return KVM_ERR_ILLEGAL_ARGUMENT;
    for (size_t i = 0; i < merged.size(); i++) {
        left[i] = (int32_t)merged[i].l;
        right[i] = (int32_t)merged[i].r;
    }
    return (int64_t)merged.size();
    });
}

extern "C" int kvm_write_series_file(const char *path, const double *values, int64_t n) {
    return kvm_guard<int>(nullptr, [&]() -> int {
    if (!path || !values || n < 0) return KVM_ERR_ILLEGAL_ARGUMENT;
    FILE *f = std::fopen(path, "wb");
    if (!f) return KVM_ERR_IO;
    std::vector<unsigned char> buf(1 << 20);
    int64_t i = 0;
    while (i < n) {
        size_t cnt = (size_t)std::min<int64_t>((int64_t)buf.size() / 8, n - i);
        for (size_t c = 0; c < cnt; c++) {
            uint64_t v;
            std::memcpy(&v, &values[i + (int64_t)c], 8);
            v = __builtin_bswap64(v);  // DataOutputStream.writeDouble is big-endian (K/DataGenerator.java:111)
            std::memcpy(&buf[c * 8], &v, 8);
        }
        if (std::fwrite(buf.data(), 8, cnt, f) != cnt) {
            std::fclose(f);
            return KVM_ERR_IO;
        }
        i += (int64_t)cnt;
    }
    std::fclose(f);
    return KVM_OK;
    });
}
