// The latency-bound streams without Python: T host threads, one context each over one resident series (kvm_share_series),
// RSM-ED with the reference's early abandoning / cNSM-ED on the bench list (n = 1e8, m = 8192, 1e6 candidates, 10 queries
// x 4 lists).  Separates the library's own ceiling from the Python driver's (GIL).
// Build: g++ -O2 -std=c++17 stream_native.cpp -o stream_native -I../../include -L../../kv-match_amd/csrc -lkvmatch_hip \
//        -Wl,-rpath,'$ORIGIN/../../kv-match_amd/csrc' -lpthread        Run: ./stream_native [calls] [variant: 0 rsm, 1 cnsm]
#include "kvmatch.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int calls = argc > 1 ? atoi(argv[1]) : 400, variant = argc > 2 ? atoi(argv[2]) : 0;
    const int64_t n = 100000000; const int32_t m = 8192;
    std::vector<double> s((size_t)n);
    kvm_gen_series(0x4B564D31ULL, n, s.data());
    std::mt19937_64 rng(5);
    std::vector<int32_t> qoffs;
    for (int i = 0; i < 10; i++) qoffs.push_back((int32_t)((rng() % ((n - m) / (4 * m))) * 4 * m + 1));
    std::vector<std::vector<int32_t>> L(4), R(4);
    for (int j = 0; j < 4; j++) {
        L[j].resize(200000); R[j].resize(200000);
        const int64_t k = kvm_gen_intervals(11 + j, n, m, 1000000, 64.0, qoffs.data(), 10, L[j].data(), R[j].data(), 200000);
        L[j].resize(k); R[j].resize(k);
    }
    kvm_ctx *own; if (kvm_open(0, &own)) { fprintf(stderr, "open: %s\n", kvm_last_error(nullptr)); return 1; }
    if (kvm_load_series_f64(own, s.data(), n)) { fprintf(stderr, "load: %s\n", kvm_last_error(own)); return 1; }
    for (int T : {1, 2, 3, 5, 8, 12, 16}) {
        std::vector<kvm_ctx *> ctx{own};
        for (int t = 1; t < T; t++) { kvm_ctx *c; kvm_open(0, &c); kvm_share_series(c, own); ctx.push_back(c); }
        for (int rep = 0; rep < 2; rep++) {
            std::atomic<int> next{0}; std::atomic<long long> cand{0}, answers{0};
            const int total = rep == 0 ? 60 : calls;
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&, t] {
                for (;;) {
                    const int i = next.fetch_add(1); if (i >= total) return;
                    const double *q = s.data() + qoffs[i % 10] - 1; kvm_result *r = nullptr;
                    const auto &l = L[i % 4], &rr = R[i % 4];
                    const int rc = variant == 0 ? kvm_verify_rsm_ed(ctx[t], q, m, l.data(), rr.data(), (int32_t)l.size(), 0, 10.0, &r)
                                                : kvm_verify_cnsm_ed(ctx[t], q, m, l.data(), rr.data(), (int32_t)l.size(), 0, 10.0, 2.0, 10.0, &r);
                    if (rc) { fprintf(stderr, "verify: %s\n", kvm_last_error(ctx[t])); exit(1); }
                    cand += kvm_result_candidates(r); answers += kvm_result_count(r); kvm_result_free(r);
                }
            });
            for (auto &x : th) x.join();
            const double dt = now() - t0;
            if (rep) printf("%s abandon, native driver: in flight %d: %7.1f us per call, %.3e candidates/s, answers %lld\n", variant ? "cnsm_ed" : "rsm_ed", T,
                            dt / total * 1e6, cand.load() / dt, answers.load());
            fflush(stdout);
        }
        for (size_t t = 1; t < ctx.size(); t++) kvm_close(ctx[t]);
    }
    kvm_close(own);
    return 0;
}
