// kvm_cli.cpp -- minimal native driver over kvmatch_engines.hpp: the call sequence a Java host would make, for a
// box without a JDK.  Generates the seeded synthetic series, plants the query, verifies with all four engines and
// prints "Best: <offset>, distance: <d>" like the reference's log line (K/QueryEngine.java:376).
//   kvm_cli [n=1000000] [m=1024] [epsilon=10]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kvmatch_engines.hpp"

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int m = argc > 2 ? atoi(argv[2]) : 1024;
    const double eps = argc > 3 ? atof(argv[3]) : 10.0;
    try {
        std::vector<double> series((size_t)n);
        if (kvm_gen_series(0x4B564D31ULL, n, series.data()) != KVM_OK) return 2;
        const int32_t qoff = (int32_t)(n / 3 + 17);
        std::vector<double> q(series.begin() + (qoff - 1), series.begin() + (qoff - 1 + m));
        std::vector<int32_t> L((size_t)n / 16 + 16), R(L.size());
        int64_t k = kvm_gen_intervals(1, n, m, n / 100, 64.0, &qoff, 1, L.data(), R.data(), (int64_t)L.size());
        if (k < 0) return 3;
        std::vector<kvmatch::Interval> iv((size_t)k);
        for (int64_t i = 0; i < k; i++) iv[(size_t)i] = {L[(size_t)i], R[(size_t)i]};
        kvmatch::Phase2Context ctx(0);
        ctx.loadSeries(series);
        auto show = [&](const char *name, const kvmatch::Answers &a) {
            if (!a.answers.empty())
                printf("%s Best: %d, distance: %g | T_2: %.3f ms, #candidates: %lld, #answers: %zu\n", name,
                       a.answers[0].first, a.answers[0].second, a.t2_ms, (long long)a.cntCandidate, a.answers.size());
            else
                printf("%s No result for the query | #candidates: %lld\n", name, (long long)a.cntCandidate);
        };
        show("RSM-ED  ", kvmatch::QueryEngine(ctx).verify(iv, 1, q, eps));
        show("RSM-DTW ", kvmatch::QueryEngineDtw(ctx).verify(iv, 1, q, eps, 0.05));
        show("cNSM-ED ", kvmatch::NormQueryEngine(ctx).verify(iv, 1, q, eps, 1.5, 5.0));
        show("cNSM-DTW", kvmatch::NormQueryEngineDtw(ctx).verify(iv, 1, q, eps, 0.05, 1.5, 5.0));
        {   // several queries in flight over the one resident series: native host threads (kvm_run_jobs)
            kvmatch::QueryStream qs(ctx, 3);
            std::vector<int32_t> l, r;
            kvmatch::detail::flatten(iv, l, r);
            std::vector<kvm_job> jobs;
            for (int i = 0; i < 6; i++) jobs.push_back(kvmatch::QueryStream::job(i % 2, q, l, r, 1, eps, 0, 1.5, 5.0));
            auto res = qs.run(jobs);
            show("stream[0] RSM-ED ", res[0]);
            show("stream[5] cNSM-ED", res[5]);
        }
        // index-free full scans (experiments/ucr/*), step 1 of the index build, the phase-2 time model
        show("UCR-ED  ", kvmatch::UcrEdQueryExecutor(ctx, n, q, eps, 1.5, 5.0, 50000).run());
        show("UCR-DTW ", kvmatch::UcrDtwQueryExecutor(ctx, n, q, eps, kvm_resolve_rho(0.05, m), 1.5, 5.0).run());
        int64_t intervals = 0, covered = 0;
        ctx.indexRows(50, [&](double, int32_t first, int32_t last) { intervals++; covered += last - first + 1; });
        int64_t rows = 0, grouped = 0;
        double prev_key = 0;
        bool descending = true;
        ctx.indexRowsGrouped(50, [&](double key, const int32_t *, const int32_t *, int32_t count) {
            if (rows > 0 && !(key < prev_key)) descending = false;
            prev_key = key;
            rows++;
            grouped += count;
        });
        printf("index w=50: %lld intervals covering %lld positions, %lld rows (%s), grouped intervals %lld\n", (long long)intervals,
               (long long)covered, (long long)rows, descending ? "keys descending" : "KEYS NOT DESCENDING", (long long)grouped);
        printf("T_2 model (rsm_ed, %lld intervals, %lld offsets, m=%d): %.4f ms\n", (long long)k, (long long)(n / 100), m,
               kvm_phase2_time_estimate_ms(0, k, n / 100, m));
    } catch (const kvmatch::IllegalArgumentException &e) {
        fprintf(stderr, "IllegalArgumentException: %s\n", e.what());
        return 1;
    } catch (const kvmatch::IOException &e) {
        fprintf(stderr, "IOException: %s\n", e.what());
        return 1;
    }
    return 0;
}
