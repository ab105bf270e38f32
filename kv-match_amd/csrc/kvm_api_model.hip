// kvm_api_model.hip -- host-only entry points: the phase-2 time model and the early-termination test built on it
// (K/QueryEngine.java:55-57,316-327), kvm_resolve_rho, kvm_cnsm_dtw_order (JDK 8 TimSort restated, kvm_jsort.h)
#include "kvm_api_internal.h"
#include "kvm_jsort.h"

// The reference's phase-2 time model T_2 = A * #intervals + B * (#offsets / 1e5) * m (K/QueryEngine.java:55-57,316-327) with
// the terms this path adds: C, the launch chain of a call, and D, the one unit of work that holds an answer (a query drawn
// from the series always has its self match) and walks the whole window -- m points for the ED engines, one banded DP of
// m macro-steps x O(m) cells (rho = 5 percent of m) for the DTW engines -- and E, the part of a candidate's cost that does not
// grow with m.  The constants are generated: profiles/fit_cost_model.py writes kvm_cost_model.inc next to
// profiles/r04_phase2_cost_model.json (non-negative least squares on relative residuals, 27-48 (intervals, offsets, m) points
// per variant on n = 1e8, early abandon on, selective queries, one MI355X): {C [ms], A [ms per interval], B [ms per 1e5
// offset-points], D, E [ms per 1e5 offsets]}; tests/test_full_size_gpu.py holds the estimate to measured calls.
extern "C" double kvm_phase2_time_estimate_ms(int32_t variant, int64_t disjoint_windows, int64_t candidate_offsets, int32_t length) {
    static const double K[4][5] = {
#include "kvm_cost_model.inc"
    };
    if (variant < 0 || variant > 3) return -1.0;
    const double m = (double)length;
    const double floor_units = variant >= 2 ? m * m / 1e3 : m;
    return K[variant][0] + K[variant][1] * (double)disjoint_windows +
           (K[variant][2] * m + K[variant][4]) * ((double)candidate_offsets / 1e5) + K[variant][3] * floor_units;
}

// The same estimate when the number of candidates that reach the banded DP is known (the library knows it 0.1 ms into a DTW
// call, behind the screens; KVM_STAT_PASS_LB_KEOGH2 of a finished call): the D term above stands for ONE DP -- the lone wave of
// a selective query -- and the DPs of a call run in rounds of the resident waves (three per SIMD: 12 per CU), a last round of
// <= 4 / <= 8 per CU at 0.46 / 0.65 of a full one (kvm_dtw_dp.h).  `dp_count` < 0, or an ED variant: the a-priori form.  With
// the reference's early abandoning the DPs of a long list end early, so for such lists this is an upper bound.
extern "C" double kvm_phase2_time_estimate_dp_ms(int32_t variant, int64_t disjoint_windows, int64_t candidate_offsets, int32_t length,
                                                 int64_t dp_count, int32_t num_cu) {
    const double base = kvm_phase2_time_estimate_ms(variant, disjoint_windows, candidate_offsets, length);
    if (base < 0 || variant < 2 || dp_count < 0) return base;
    static const double K[4][5] = {
#include "kvm_cost_model.inc"
    };
    const double m = (double)length, one_dp = K[variant][3] * m * m / 1e3;
    const int64_t round = 12LL * (num_cu > 0 ? num_cu : 256), tail = dp_count % round;
    double units = (double)(dp_count / round) / 0.46;   // full rounds, in units of a lone DP
    if (tail) units += 3 * tail <= round ? 1.0 : (3 * tail <= 2 * round ? 0.65 / 0.46 : 1.0 / 0.46);
    return base - one_dp + one_dp * units;   // (no DP at all: the call is its chain and its screens)
}

// The phase-1 early-termination test of the four engines (K/QueryEngine.java:316-327: `i >= 5 && totalTimeUsageEstimated >
// lastTotalTimeUsageEstimated`), deterministic: the caller supplies its phase-1 cost so far from a model of its own (index rows
// scanned x a constant, say) instead of reading the wall clock, the phase-2 term is the estimate above.
extern "C" int32_t kvm_phase1_should_stop(int32_t variant, int32_t segment_index, double phase1_ms_so_far, int64_t disjoint_windows,
                                          int64_t candidate_offsets, int32_t length, double *last_total_ms) {
    if (!last_total_ms) return -1;
    const double est = kvm_phase2_time_estimate_ms(variant, disjoint_windows, candidate_offsets, length);
    if (est < 0) return -1;
    const double total = phase1_ms_so_far + est;
    if (segment_index >= 5 && total > *last_total_ms) return 1;   // (`break`: lastTotalTimeUsageEstimated is not updated, :321-324)
    *last_total_ms = total;                                        // :325
    return 0;
}

extern "C" int32_t kvm_resolve_rho(double r, int32_t m) {
    if (r <= 1) return (int32_t)std::floor(r * m);  // K/QueryEngineDtw.java:135-140
    return (int32_t)std::floor(r);
}

extern "C" int kvm_cnsm_dtw_order(const double *query, int32_t m, int32_t *order) {
    return kvm_guard<int>(nullptr, [&]() -> int {
    if (!query || m < 1) return kvm_fail(nullptr, KVM_ERR_ILLEGAL_ARGUMENT, "kvm_cnsm_dtw_order: bad arguments");
    // K/NormQueryEngineDtw.java:205-211 (statistics) and :461-464 (z-normalisation)
    double ex = 0, ex2 = 0;
    for (int i = 0; i < m; i++) {
        ex += query[i];
        ex2 += query[i] * query[i];
    }
    const double mean = ex / m, sd = std::sqrt(ex2 / m - mean * mean);
    std::vector<double> zq((size_t)m);
    for (int i = 0; i < m; i++) zq[(size_t)i] = (query[i] - mean) / sd;
    std::vector<int32_t> ord((size_t)m);
    const bool ok = kvm_jsort::cnsm_dtw_order(zq.data(), m, ord.data());
    if (order) std::memcpy(order, ord.data(), (size_t)m * sizeof(int32_t));
    return ok ? KVM_OK : kvm_fail(nullptr, KVM_ERR_ILLEGAL_ARGUMENT, "Comparison method violates its general contract!");
    });
}

