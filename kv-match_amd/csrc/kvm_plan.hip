// kvm_plan.hip -- the plan of a call: the phase-1 interval list -> totals the launch geometry and the buffers need, the first
// offender the reference would have met (K/QueryEngine.java:343-352, K/NormQueryEngine.java:454-496), the query envelope; trace switches
#include "kvm_api_internal.h"

bool kvm_trace_on() {
    static int v = -1;
    if (v < 0) v = getenv("KVM_TRACE") ? 1 : 0;
    return v == 1;
}
// KVM_DEBUG_SYNC=1 in the environment: wait for the stream after every stage of a DTW call and name it on stderr (finds a
// kernel that does not return)
bool kvm_debug_sync_on() {
    static int v = -1;
    if (v < 0) v = getenv("KVM_DEBUG_SYNC") ? 1 : 0;
    return v == 1;
}

// ---- plan: intervals -> runs -> tiles -------------------------------------------------------------------


// Totals of the plan over the interval list (the per-run prefix sums are formed on the device from the same rule,
// kvm_plan_interval).  Written as a plain reduction so that the compiler vectorises it; the AVX2 clone is picked at
// run time (15 675 intervals: 31 us scalar, a few us vectorised).
struct plan_totals {
    int64_t cnt_candidate = 0, n_tiles = 0, n_tiles_b = 0, n_tiles_c = 0, windows = 0, nr = 0, bad_rsm = 0;
};
#define PLAN_TOTALS_BODY                                                                                        \
    int64_t cnt = 0, ta = 0, tb = 0, tc = 0, w = 0, nr = 0, bad = 0;                                            \
    for (int32_t p = 0; p < k; p++) {                                                                           \
        const int64_t l = left[p], r = right[p];                                                                \
        cnt += r - l + 1;                                                                                       \
        int64_t lo = l - shift, hi = r - shift;                                                                 \
        lo = lo < 1 ? 1 : lo;                                                                                   \
        hi = hi > hi_max ? hi_max : hi;                                                                         \
        const int64_t c64 = hi - lo + 1;                                                                        \
        bad |= (c64 + m - 1 < 1) ? 1 : 0;                                                                       \
        const int64_t c = c64 > 0 ? c64 : 0;                                                                    \
        const int64_t rem = c & 63;                                                                             \
        const int64_t a_h = (c >> 6) + (rem >= (int64_t)KVM_EDA_MIN ? 1 : 0);                                   \
        const int64_t b_h = (rem > (int64_t)KVM_EDC_TILE && rem < (int64_t)KVM_EDA_MIN) ? 1 : 0;                \
        const int64_t c_h = (rem >= 1 && (rem <= (int64_t)KVM_EDC_TILE ||                                       \
                                          (rem > (int64_t)KVM_EDB_TILE && rem < (int64_t)KVM_EDA_MIN))) ? 1 : 0; \
        const int64_t a_u = (c + ((int64_t)1 << tile_log2) - 1) >> tile_log2;                                   \
        ta += hybrid ? a_h : a_u;                                                                               \
        tb += hybrid ? b_h : 0;                                                                                 \
        tc += hybrid ? c_h : 0;                                                                                 \
        w += c;                                                                                                 \
        nr += c > 0 ? 1 : 0;                                                                                    \
    }                                                                                                           \
    out.cnt_candidate = cnt; out.n_tiles = ta; out.n_tiles_b = tb; out.n_tiles_c = tc; out.windows = w; out.nr = nr; out.bad_rsm = bad;
__attribute__((target("avx2"))) static void plan_totals_avx2(const int32_t *left, const int32_t *right, int32_t k, int64_t shift,
                                                             int64_t hi_max, int64_t m, bool hybrid, int tile_log2,
                                                             plan_totals &out) {
    PLAN_TOTALS_BODY
}
static void plan_totals_generic(const int32_t *left, const int32_t *right, int32_t k, int64_t shift, int64_t hi_max, int64_t m,
                                bool hybrid, int tile_log2, plan_totals &out) {
    PLAN_TOTALS_BODY
}
#undef PLAN_TOTALS_BODY

// Window start offsets of interval [left,right]: max(1,left-shift) .. min(right-shift, n-m+1)
// (K/QueryEngine.java:346-354: begin/end clipping, loop `i + length - 1 < data.size()`).
// `rho_strict` >= 0 (DTW engines with KVM_OPT_STRICT_BOUNDS): a fetched range of <= rho_strict points ends the reference in
// ArrayIndexOutOfBoundsException (lowerUpperLemire over the fetched data, K/QueryEngineDtw.java:399,
// K/NormQueryEngineDtw.java:524 -> K/utils/DtwUtils.java:81-90) -- right after the readTimeSeries that could have thrown
// IllegalArgumentException for the same range, interval by interval (chunk by chunk) in list order.
int kvm_build_plan(kvm_ctx *ctx, bool cnsm, int32_t m, const int32_t *left, const int32_t *right, int32_t k,
                      int32_t shift, int32_t tile_max, bool hybrid, plan &pl, int32_t rho_strict) {
    const int64_t n = ctx->n;
    // The engines hand phase 2 the list sortAndMergeIntervals leaves (K/QueryEngine.java:664-693): sorted by `left`.
    // Lists with overlapping, adjacent, nested, repeated or empty intervals go through like in the Java loops (every
    // interval is scanned on its own: a candidate's records carry the INDEX of its run, two runs may well start at the
    // same offset).  A list that is NOT sorted by `left`: the RSM engines treat intervals independently, so it is sorted
    // here (stable, by `left`) -- the same (offset, distance) pairs come out, in the tie order of the sorted list.  The cNSM
    // engines merge consecutive intervals into one fetched chunk and index it with `left - begin`: out of order that is an
    // IndexOutOfBoundsException or the range check of readTimeSeries in Java, depending on the list; refused here.
    std::vector<int32_t> sorted_l, sorted_r, clipped_r;
    const int32_t *const left_given = left, *const right_given = right;  // (which interval throws first: the list as given)
    {
        bool ascending = true;
        for (int32_t p = 1; p < k; p++) ascending &= left[p] >= left[p - 1];
        if (!ascending) {
            if (cnsm)
                return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, "interval list not sorted by left (sortAndMergeIntervals, K/QueryEngine.java:664-693)");
            std::vector<int32_t> idx((size_t)k);
            for (int32_t p = 0; p < k; p++) idx[(size_t)p] = p;
            std::stable_sort(idx.begin(), idx.end(), [left](int32_t a, int32_t b) { return left[a] < left[b]; });
            sorted_l.resize((size_t)k);
            sorted_r.resize((size_t)k);
            for (int32_t p = 0; p < k; p++) {
                sorted_l[(size_t)p] = left[idx[(size_t)p]];
                sorted_r[(size_t)p] = right[idx[(size_t)p]];
            }
            left = sorted_l.data();
            right = sorted_r.data();
        }
    }
    // RSM engines fetch per interval: readTimeSeries(begin, end-begin+1) throws when the clipped range is empty
    // (K/QueryEngine.java:346-352; K/operator/file/TimeSeriesFileOperator.java:55-57).  end - begin + 1 == (clipped window
    // count) + m - 1, so the totals loop below notices an offender (and then words the message from the first one).
    if (cnsm) {
        // cNSM engines fetch greedy-merged chunks of <= MAX_SCAN_DATA_LENGTH points
        // (K/NormQueryEngine.java:454-482); only the chunk read can throw.
        int32_t idx = 0;
        while (idx < k) {
            int32_t beginIdx = idx, endIdx = idx;
            int64_t begin = (int64_t)left[idx] - shift, end = (int64_t)right[idx] - shift + m - 1;
            if (begin < 1) begin = 1;
            int64_t length = end - begin + 1;
            idx++;
            while (idx < k) {
                begin = (int64_t)left[idx] - shift;
                int64_t newLength = length + begin - end - 1;
                end = (int64_t)right[idx] - shift + m - 1;
                if (end > n) end = n;
                newLength += end - begin + 1;
                if (newLength > KVM_MAX_SCAN_DATA_LENGTH) break;
                endIdx = idx;
                length = newLength;
                idx++;
            }
            begin = (int64_t)left[beginIdx] - shift;
            end = (int64_t)right[endIdx] - shift + m - 1;
            if (begin < 1) begin = 1;
            if (end > n) end = n;
            if (end - begin + 1 < 1 || begin > n) {
                char msg[96];
                snprintf(msg, sizeof msg, "left: %lld, length: %lld", (long long)begin, (long long)(end - begin + 1));
                return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, msg);
            }
            if (rho_strict >= 0 && end - begin + 1 <= rho_strict) {
                char msg[32];
                snprintf(msg, sizeof msg, "%lld", (long long)(end - begin + 1 - rho_strict - 1));  // (JDK 8 words it as the index)
                return kvm_fail(ctx, KVM_ERR_INDEX_OUT_OF_BOUNDS, msg);
            }
            if (pl.chunk_lo)  // the data envelope of the DTW cascade is built per fetched chunk (K/NormQueryEngineDtw.java:522-524)
                for (int32_t p = beginIdx; p <= endIdx; p++) {
                    pl.chunk_lo[p] = (int32_t)begin;
                    pl.chunk_hi[p] = (int32_t)end;
                }
            // An interval is scanned inside its chunk only: `if (end1 > data.size() - 1) end1 = data.size() - 1`
            // (K/NormQueryEngine.java:494, K/NormQueryEngineDtw.java:536).  The chunk ends where its LAST interval ends, so
            // in a list with overlapping or nested intervals an earlier, longer interval loses the windows that reach past
            // it (found by profiles/fuzz_lists.py); with disjoint sorted intervals the clip never bites.
            for (int32_t p = beginIdx; p <= endIdx; p++) {
                const int64_t last_start = end - m + 1 + shift;  // in the coordinates of `right`
                if ((int64_t)right[p] > last_start) {
                    if (clipped_r.empty()) clipped_r.assign(right, right + k);
                    clipped_r[(size_t)p] = (int32_t)std::max<int64_t>(last_start, (int64_t)left[p] - 1);
                }
            }
        }
    }
    // (cntCandidate is counted from the list as given, K/NormQueryEngine.java:485; the plan from the clipped one)
    int64_t cnt_given = 0;
    const int32_t *right_plan = right;
    if (!clipped_r.empty()) {
        for (int32_t p = 0; p < k; p++) cnt_given += (int64_t)right[p] - left[p] + 1;
        right_plan = clipped_r.data();
    }
    // totals only (launch geometry, buffer sizes): the per-run prefix sums are formed on the device from the same rule
    const int64_t hi_max = n - m + 1;
    int tile_log2 = 0;
    while ((1 << tile_log2) < tile_max) tile_log2++;
    plan_totals tot;
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2) plan_totals_avx2(left, right_plan, k, shift, hi_max, m, hybrid, tile_log2, tot);
    else plan_totals_generic(left, right_plan, k, shift, hi_max, m, hybrid, tile_log2, tot);
    const int64_t cnt_candidate = clipped_r.empty() ? tot.cnt_candidate : cnt_given, n_tiles = tot.n_tiles, n_tiles_b = tot.n_tiles_b, n_tiles_c = tot.n_tiles_c, windows = tot.windows,
                  nr = tot.nr;
    const bool bad_rsm = tot.bad_rsm != 0;
    // (a stub of <= rho < m points holds no window: windows < list length is the cheap sign that one may exist)
    if ((bad_rsm || (rho_strict >= 0 && nr < k)) && !cnsm) {
        for (int32_t p = 0; p < k; p++) {
            int64_t begin = (int64_t)left_given[p] - shift, end = (int64_t)right_given[p] - shift + m - 1;
            if (begin < 1) begin = 1;
            if (end > n) end = n;
            if (end - begin + 1 < 1) {
                char msg[96];
                snprintf(msg, sizeof msg, "left: %lld, length: %lld", (long long)begin, (long long)(end - begin + 1));
                return kvm_fail(ctx, KVM_ERR_ILLEGAL_ARGUMENT, msg);
            }
            if (rho_strict >= 0 && end - begin + 1 <= rho_strict) {
                char msg[32];
                snprintf(msg, sizeof msg, "%lld", (long long)(end - begin + 1 - rho_strict - 1));
                return kvm_fail(ctx, KVM_ERR_INDEX_OUT_OF_BOUNDS, msg);
            }
        }
    }
    if (k > 0) {
        std::memcpy(pl.left, left, (size_t)k * 4);
        std::memcpy(pl.right, right_plan, (size_t)k * 4);
    }
    pl.cnt_candidate += cnt_candidate;
    pl.n_tiles += n_tiles;
    pl.n_tiles_b += n_tiles_b;
    pl.n_tiles_c += n_tiles_c;
    pl.windows += windows;
    pl.n_runs = nr;
    return KVM_OK;
}

// Upper/lower envelope of the query: u[i] = max q[max(0,i-r)..min(m-1,i+r)], l[i] = min (same values as
// DtwUtils.lowerUpperLemire, K/utils/DtwUtils.java:50-91, which is a pure selection).  Monotonic deques.
void kvm_query_envelope(const std::vector<double> &q, int r, std::vector<double> &u, std::vector<double> &l) {
    const int m = (int)q.size();
    u.assign(m, 0.0);
    l.assign(m, 0.0);
    std::deque<int> mx, mn;
    int next = 0;
    for (int i = 0; i < m; i++) {
        int hi = std::min(m - 1, i + r), lo = std::max(0, i - r);
        for (; next <= hi; next++) {
            while (!mx.empty() && q[mx.back()] <= q[next]) mx.pop_back();
            mx.push_back(next);
            while (!mn.empty() && q[mn.back()] >= q[next]) mn.pop_back();
            mn.push_back(next);
        }
        while (mx.front() < lo) mx.pop_front();
        while (mn.front() < lo) mn.pop_front();
        u[i] = q[mx.front()];
        l[i] = q[mn.front()];
    }
}

