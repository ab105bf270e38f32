// Memory-safety fuzz of the library's TimSort restatement (kv-match_amd/csrc/kvm_jsort.h) under ASan/UBSan: whatever the
// comparator answers -- an arbitrary hash, the truncating comparator on huge / NaN / Inf values -- every index stays in
// bounds and the array remains a permutation, whether the sort "throws" or not.
#include "../../kv-match_amd/csrc/kvm_jsort.h"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <algorithm>
#include <cmath>
int main() {
    std::mt19937_64 rng(12345);
    long thrown = 0, ok = 0;
    for (int trial = 0; trial < 4000; trial++) {
        int n = (int)(rng() % (trial % 50 == 0 ? 20000 : 600));
        std::vector<int32_t> a((size_t)n);
        for (int i = 0; i < n; i++) a[(size_t)i] = i;
        const uint64_t salt = rng();
        const int mode = trial % 4;
        std::vector<double> z((size_t)n);
        for (int i = 0; i < n; i++) {
            double u = (double)(rng() % 2000001) / 1e6 - 1.0;
            z[(size_t)i] = mode == 2 ? u * 1e12 : mode == 3 ? (rng() % 17 == 0 ? NAN : (rng() % 13 == 0 ? INFINITY : u * 4)) : u * 3;
        }
        bool r;
        if (mode == 0) {  // arbitrary (hash) comparator: any int, no consistency at all
            r = kvm_jsort::java_arrays_sort(a.data(), n, [salt](int32_t x, int32_t y) {
                uint64_t h = (uint64_t)x * 0x9E3779B97F4A7C15ULL ^ (uint64_t)y * 0xC2B2AE3D27D4EB4FULL ^ salt;
                h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
                return (int)(h % 5) - 2;
            });
        } else {
            r = kvm_jsort::cnsm_dtw_order(z.data(), n, a.data());
        }
        // whatever happened, the array must still be a permutation
        std::vector<int32_t> b(a);
        std::sort(b.begin(), b.end());
        for (int i = 0; i < n; i++) if (b[(size_t)i] != i) { printf("NOT A PERMUTATION trial %d\n", trial); return 1; }
        (r ? ok : thrown)++;
    }
    printf("ok %ld thrown %ld\n", ok, thrown);
    return 0;
}
