// kvmatch_engines.hpp -- C++ host-side mirror of the reference's four query engines, restricted to the phase-2
// seam (header-only RAII wrappers over the C ABI of include/kvmatch.h).  Class names, argument order and error
// behaviour follow the reference: K/QueryEngine.java, K/QueryEngineDtw.java, K/NormQueryEngine.java,
// K/NormQueryEngineDtw.java (K/ = src/main/java/cn/edu/fudan/dsm/kvmatch/).  Phases 0/1 stay in the Java host.
#pragma once
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <utility>
#include <memory>
#include <new>
#include <vector>

#include "../../include/kvmatch.h"

namespace kvmatch {

// java.lang.IllegalArgumentException (K/operator/file/TimeSeriesFileOperator.java:55-57)
struct IllegalArgumentException : std::invalid_argument { using std::invalid_argument::invalid_argument; };
// java.io.IOException (what query() declares, K/QueryEngine.java:155); device failures map here too
struct IOException : std::runtime_error { using std::runtime_error::runtime_error; };
// java.lang.ArrayIndexOutOfBoundsException: where DtwUtils.lowerUpperLemire indexes below its arrays (K/utils/DtwUtils.java:81-90)
struct ArrayIndexOutOfBoundsException : std::out_of_range { using std::out_of_range::out_of_range; };

struct Interval { int32_t left, right; };              // K/common/Interval.java:23-35 (epsilon is not read by phase 2)
using Answer = std::pair<int32_t, double>;             // Pair<Integer, Double>, K/QueryEngine.java:341
constexpr int WU0 = 25;                                // WuList[0], K/QueryEngine.java:44

struct Answers {
    std::vector<Answer> answers;   // sorted by distance (K/QueryEngine.java:373)
    int64_t cntCandidate = 0;      // statistics[3]
    double t2_ms = 0;              // statistics[2]
};

class Phase2Context {
public:
    explicit Phase2Context(int device = 0) { check(kvm_open(device, &ctx_), nullptr); }
    ~Phase2Context() { if (ctx_) kvm_close(ctx_); }
    Phase2Context(const Phase2Context &) = delete;
    Phase2Context &operator=(const Phase2Context &) = delete;

    void loadSeries(const std::vector<double> &v) { check(kvm_load_series_f64(ctx_, v.data(), (int64_t)v.size()), ctx_); }
    void loadSeriesFile(const std::string &path, int64_t n) { check(kvm_load_series_file(ctx_, path.c_str(), n), ctx_); }
    // TimeSeriesOperator.readTimeSeries(left, length), K/operator/TimeSeriesOperator.java:38
    std::vector<double> readTimeSeries(int32_t left, int32_t length) {
        std::vector<double> out(length > 0 ? (size_t)length : 0);
        check(kvm_read_time_series(ctx_, left, length, out.data()), ctx_);
        return out;
    }
    // Step 1 of IndexBuilder.buildIndexes for window length w (K/IndexBuilder.java:186-300): intervals of consecutive
    // window positions with the same row key, in position order; `fn(key, first, last)` is called once per interval
    // (the host groups them into IndexNode rows, :270-284, and runs step 2).
    template <class Fn>
    void indexRows(int32_t w, Fn &&fn) {
        kvm_index_rows *r = nullptr;
        check(kvm_index_rows_build(ctx_, w, &r), ctx_);
        const int64_t cnt = kvm_index_rows_count(r);
        const double *keys = kvm_index_rows_keys(r);
        const int32_t *first = kvm_index_rows_first(r), *last = kvm_index_rows_last(r);
        try {
            for (int64_t i = 0; i < cnt; i++) fn(keys[i], first[i], last[i]);
        } catch (...) {
            kvm_index_rows_free(r);
            throw;
        }
        kvm_index_rows_free(r);
    }
    // The same intervals collected into rows: `row(key, first, last, count)` once per distinct key, in the order of
    // step 2 (descending key, K/IndexBuilder.java:318), with pointers to the row's `count` intervals in position order.
    template <class Fn>
    void indexRowsGrouped(int32_t w, Fn &&row) {
        kvm_index_rows *r = nullptr;
        check(kvm_index_rows_build_grouped(ctx_, w, &r), ctx_);
        const int64_t ng = kvm_index_rows_groups(r);
        const double *gk = kvm_index_rows_group_keys(r);
        const int32_t *go = kvm_index_rows_group_offsets(r);
        const int32_t *first = kvm_index_rows_first(r), *last = kvm_index_rows_last(r);
        try {
            for (int64_t g = 0; g < ng; g++) row(gk[g], first + go[g], last + go[g], go[g + 1] - go[g]);
        } catch (...) {
            kvm_index_rows_free(r);
            throw;
        }
        kvm_index_rows_free(r);
    }
    // IndexBuilder.SingleIndexBuilder.run + IndexFileOperator.writeAll for window length w (K/IndexBuilder.java:186-347,
    // K/operator/file/IndexFileOperator.java:127-165): the bytes of files/index-{N}-{Wu}; `path` non-empty also writes them
    std::vector<uint8_t> indexBuild(int32_t w, const std::string &path = std::string()) {
        kvm_index_file *f = nullptr;
        check(kvm_index_build(ctx_, w, &f), ctx_);
        const uint8_t *p = kvm_index_file_bytes(f);
        std::vector<uint8_t> image(p, p + kvm_index_file_size(f));
        const int rc = path.empty() ? KVM_OK : kvm_index_file_write(f, path.c_str());
        kvm_index_file_free(f);
        if (rc) throw IOException("cannot write " + path);
        return image;
    }
    // one more engine over the series `owner` has staged (kvm_share_series): a second query in flight
    void shareSeries(const Phase2Context &owner) { check(kvm_share_series(ctx_, owner.ctx_), ctx_); }
    void setOption(int32_t option, int64_t value) { check(kvm_set_option(ctx_, option, value), ctx_); }
    int64_t seriesLength() const { return kvm_series_length(ctx_); }
    kvm_ctx *handle() const { return ctx_; }

    static void check(int rc, const kvm_ctx *ctx) {
        if (rc == KVM_OK) return;
        std::string msg = kvm_last_error(ctx);
        if (rc == KVM_ERR_ILLEGAL_ARGUMENT) throw IllegalArgumentException(msg);
        if (rc == KVM_ERR_INDEX_OUT_OF_BOUNDS) throw ArrayIndexOutOfBoundsException(msg);  // (KVM_OPT_STRICT_BOUNDS)
        if (rc == KVM_ERR_NO_MEMORY) throw std::bad_alloc();                                // (OutOfMemoryError in the JNI shim)
        throw IOException("[" + std::to_string(rc) + "] " + msg);
    }

private:
    kvm_ctx *ctx_ = nullptr;
};

namespace detail {
inline void flatten(const std::vector<Interval> &v, std::vector<int32_t> &l, std::vector<int32_t> &r) {
    l.resize(v.size());
    r.resize(v.size());
    for (size_t i = 0; i < v.size(); i++) { l[i] = v[i].left; r[i] = v[i].right; }
}
inline Answers collect(kvm_result *res) {
    Answers a;
    const int64_t cnt = kvm_result_count(res);
    const int32_t *off = kvm_result_offsets(res);
    const double *dist = kvm_result_distances(res);
    a.answers.reserve((size_t)cnt);
    for (int64_t i = 0; i < cnt; i++) a.answers.emplace_back(off[i], dist[i]);
    a.cntCandidate = kvm_result_candidates(res);
    a.t2_ms = kvm_result_elapsed_ms(res);
    kvm_result_free(res);
    return a;
}
}  // namespace detail

// Several queries in flight: n contexts over one resident series, one native host thread each (kvm_run_jobs) -- the loop a
// server runs with one engine instance per thread (the engines are not re-entrant, K/QueryEngine.java:61-68).  Jobs are the
// arguments of the engines' verify() as data; the answers come back in job order.
class QueryStream {
public:
    QueryStream(Phase2Context &owner, int in_flight) {
        ctxs_.push_back(owner.handle());
        for (int i = 1; i < in_flight; i++) {
            extra_.emplace_back(new Phase2Context(0));
            extra_.back()->shareSeries(owner);
            ctxs_.push_back(extra_.back()->handle());
        }
    }
    static kvm_job job(int32_t variant, const std::vector<double> &q, const std::vector<int32_t> &left, const std::vector<int32_t> &right,
                       int lastSegment, double epsilon, int32_t rho = 0, double alpha = 0, double beta = 0) {
        kvm_job j{};
        j.variant = variant; j.m = (int32_t)q.size(); j.query = q.data(); j.left = left.data(); j.right = right.data();
        j.k = (int32_t)left.size(); j.shift = (lastSegment - 1) * 25; j.epsilon = epsilon; j.rho = rho; j.alpha = alpha; j.beta = beta;
        return j;
    }
    std::vector<Answers> run(const std::vector<kvm_job> &jobs) {
        std::vector<kvm_result *> res(jobs.size(), nullptr);
        std::vector<int32_t> st(jobs.size(), 0);
        const int rc = kvm_run_jobs(ctxs_.data(), (int32_t)ctxs_.size(), jobs.data(), (int64_t)jobs.size(), res.data(), st.data());
        if (rc != KVM_OK) {
            for (kvm_result *r : res) kvm_result_free(r);
            Phase2Context::check(rc, ctxs_[0]);   // (kvm_run_jobs leaves the first failing job's message in the first context)
        }
        std::vector<Answers> out;
        out.reserve(res.size());
        for (kvm_result *r : res) out.push_back(detail::collect(r));
        return out;
    }

private:
    std::vector<kvm_ctx *> ctxs_;
    std::vector<std::unique_ptr<Phase2Context>> extra_;
};

// RSM-ED: verify() replaces K/QueryEngine.java:338-363 (+ sort :373)
class QueryEngine {
public:
    explicit QueryEngine(Phase2Context &c) : c_(c) {}
    Answers verify(const std::vector<Interval> &validPositions, int lastSegment, const std::vector<double> &queryData,
                   double epsilon) {
        std::vector<int32_t> l, r;
        detail::flatten(validPositions, l, r);
        kvm_result *res = nullptr;
        Phase2Context::check(kvm_verify_rsm_ed(c_.handle(), queryData.data(), (int32_t)queryData.size(), l.data(),
                                               r.data(), (int32_t)l.size(), (lastSegment - 1) * WU0, epsilon, &res),
                             c_.handle());
        return detail::collect(res);
    }
private:
    Phase2Context &c_;
};

// RSM-DTW: verify() replaces K/QueryEngineDtw.java:349-452; rho <= 1 is a fraction of the query length (:135-140)
class QueryEngineDtw {
public:
    explicit QueryEngineDtw(Phase2Context &c) : c_(c) {}
    Answers verify(const std::vector<Interval> &validPositions, int lastSegment, const std::vector<double> &queryData,
                   double epsilon, double rho) {
        std::vector<int32_t> l, r;
        detail::flatten(validPositions, l, r);
        kvm_result *res = nullptr;
        const int32_t m = (int32_t)queryData.size();
        Phase2Context::check(kvm_verify_rsm_dtw(c_.handle(), queryData.data(), m, l.data(), r.data(), (int32_t)l.size(),
                                                (lastSegment - 1) * WU0, epsilon, kvm_resolve_rho(rho, m), &res),
                             c_.handle());
        return detail::collect(res);
    }
private:
    Phase2Context &c_;
};

// cNSM-ED: verify() replaces K/NormQueryEngine.java:432-528
class NormQueryEngine {
public:
    explicit NormQueryEngine(Phase2Context &c) : c_(c) {}
    Answers verify(const std::vector<Interval> &validPositions, int lastSegment, const std::vector<double> &queryData,
                   double epsilon, double alpha, double beta) {
        std::vector<int32_t> l, r;
        detail::flatten(validPositions, l, r);
        kvm_result *res = nullptr;
        Phase2Context::check(kvm_verify_cnsm_ed(c_.handle(), queryData.data(), (int32_t)queryData.size(), l.data(),
                                                r.data(), (int32_t)l.size(), (lastSegment - 1) * WU0, epsilon, alpha,
                                                beta, &res),
                             c_.handle());
        return detail::collect(res);
    }
private:
    Phase2Context &c_;
};

// cNSM-DTW: verify() replaces K/NormQueryEngineDtw.java:457-603
class NormQueryEngineDtw {
public:
    explicit NormQueryEngineDtw(Phase2Context &c) : c_(c) {}
    Answers verify(const std::vector<Interval> &validPositions, int lastSegment, const std::vector<double> &queryData,
                   double epsilon, double rho, double alpha, double beta) {
        std::vector<int32_t> l, r;
        detail::flatten(validPositions, l, r);
        kvm_result *res = nullptr;
        const int32_t m = (int32_t)queryData.size();
        Phase2Context::check(kvm_verify_cnsm_dtw(c_.handle(), queryData.data(), m, l.data(), r.data(),
                                                 (int32_t)l.size(), (lastSegment - 1) * WU0, epsilon,
                                                 kvm_resolve_rho(rho, m), alpha, beta, &res),
                             c_.handle());
        return detail::collect(res);
    }
private:
    Phase2Context &c_;
};

// Index-free full scans (experiments/ucr/UcrEdQueryExecutor.java, UcrDtwQueryExecutor.java): the same phase-2 path on a
// candidate list that covers the series; see kvmatch/ucr.py for how the cut of the list reproduces each executor.
namespace detail {
// KVM_OPT_UCR_LOOPS for the duration of one executor run (the engines on the same context keep the engines' loops)
struct UcrLoops {
    explicit UcrLoops(Phase2Context &c) : c_(c), old_(kvm_get_option(c.handle(), KVM_OPT_UCR_LOOPS)) { c_.setOption(KVM_OPT_UCR_LOOPS, 1); }
    ~UcrLoops() { kvm_set_option(c_.handle(), KVM_OPT_UCR_LOOPS, old_); }
    Phase2Context &c_;
    int64_t old_;
};
inline void segments(int64_t first, int64_t last, int64_t length, std::vector<int32_t> &l, std::vector<int32_t> &r) {
    for (int64_t f = first; f <= last; f += length) {
        l.push_back((int32_t)f);
        r.push_back((int32_t)std::min(f + length - 1, last));
    }
}
}  // namespace detail

// run(): answers sorted by distance, 1-based offsets (UcrEdQueryExecutor.java:161).  segment == 0: one running-sum chain
// over the whole series as the reference keeps it (sequential); segment > 0: sums restart every `segment` offsets.
class UcrEdQueryExecutor {
public:
    UcrEdQueryExecutor(Phase2Context &c, int64_t N, const std::vector<double> &queryData, double Epsilon, double Alpha,
                       double Beta, int64_t segment = 0)
        : c_(c), N_(N), q_(queryData), eps_(Epsilon), alpha_(Alpha), beta_(Beta), segment_(segment) {}
    Answers run() {
        const int64_t M = (int64_t)q_.size(), last = N_ - M + 1;
        if (M < 1 || N_ < M || N_ > kvm_series_length(c_.handle())) throw IllegalArgumentException("N outside [M, series length]");
        std::vector<int32_t> l, r;
        detail::segments(1, last, segment_ > 0 ? segment_ : last, l, r);
        // distance() of the executor abandons on `sum < bsf` (U/UcrEdQueryExecutor.java:93) where the engines' loops run on
        // `dist <= bsf`: a partial sum exactly on Epsilon^2 is an answer at distance Epsilon, and with Epsilon^2 == 0 every window
        // that passes the (Alpha, Beta) test is one at distance 0 -- reproduced by the library behind KVM_OPT_UCR_LOOPS
        detail::UcrLoops guard(c_);
        kvm_result *res = nullptr;
        Phase2Context::check(kvm_verify_cnsm_ed(c_.handle(), q_.data(), (int32_t)M, l.data(), r.data(), (int32_t)l.size(), 0, eps_,
                                                alpha_, beta_, &res),
                             c_.handle());
        Answers a = detail::collect(res);
        return a;
    }
private:
    Phase2Context &c_;
    int64_t N_;
    std::vector<double> q_;
    double eps_, alpha_, beta_;
    int64_t segment_;
};

// run(): answers sorted by distance, 0-BASED offsets as the reference reports them (UcrDtwQueryExecutor.java:283);
// EPOCH = 100000 buffers (:97) == intervals of EPOCH - M + 1 offsets.  Rho is the band half-width in points.
// The executor never clears cb1 / cb2 between windows (:157-161): at an exact LB_Keogh tie its DP reads what earlier windows
// left there and can reject a window the engines report at distance Epsilon -- reproduced behind KVM_OPT_UCR_LOOPS.
class UcrDtwQueryExecutor {
public:
    static constexpr int64_t EPOCH = 100000;
    UcrDtwQueryExecutor(Phase2Context &c, int64_t N, const std::vector<double> &queryData, double Epsilon, int32_t Rho,
                        double Alpha, double Beta)
        : c_(c), N_(N), q_(queryData), eps_(Epsilon), rho_(Rho), alpha_(Alpha), beta_(Beta) {}
    Answers run() {
        const int64_t M = (int64_t)q_.size();
        if (M < 1 || M > EPOCH || N_ < M || N_ > kvm_series_length(c_.handle()))
            throw IllegalArgumentException("N outside [M, series length] or M > EPOCH");
        std::vector<int32_t> l, r;
        detail::segments(1, N_ - M + 1, EPOCH - M + 1, l, r);
        detail::UcrLoops guard(c_);
        kvm_result *res = nullptr;
        Phase2Context::check(kvm_verify_cnsm_dtw(c_.handle(), q_.data(), (int32_t)M, l.data(), r.data(), (int32_t)l.size(), 0,
                                                 eps_, rho_, alpha_, beta_, &res), c_.handle());
        Answers a = detail::collect(res);
        for (Answer &x : a.answers) x.first -= 1;
        return a;
    }
private:
    Phase2Context &c_;
    int64_t N_;
    std::vector<double> q_;
    double eps_;
    int32_t rho_;
    double alpha_, beta_;
};

// K/IndexBuilder.java for the "file" storage type: buildIndexes() writes files/index-{n}-{Wu} for the enabled window
// lengths (WuList / WuEnabled, :52-53); a window length the series is too short for is skipped (the reference's loop logs
// the exception and carries on, :108-110).
class IndexBuilder {
public:
    IndexBuilder(Phase2Context &ctx, int64_t n) : ctx_(ctx), n_(n) {
        if (n != ctx.seriesLength()) throw IllegalArgumentException("n differs from the length of the staged series");
    }
    std::vector<int> buildIndexes(const std::string &directory = "files") {
        static const int WuList[] = {25, 50, 75, 100, 125, 150, 175, 200, 225, 250, 275, 300, 325, 350, 375, 400};
        static const bool WuEnabled[] = {true, true, false, true, false, false, false, true, false, false, false, false, false, false, false, true};
        std::vector<int> done;
        for (size_t i = 0; i < sizeof WuList / sizeof WuList[0]; i++) {
            if (!WuEnabled[i]) continue;
            try {
                ctx_.indexBuild(WuList[i], directory + "/index-" + std::to_string(n_) + "-" + std::to_string(WuList[i]));
                done.push_back(WuList[i]);
            } catch (const IllegalArgumentException &) {
            }
        }
        return done;
    }

private:
    Phase2Context &ctx_;
    int64_t n_;
};

}  // namespace kvmatch
