/*
 * kvmatch.h -- C ABI of the MI355X-native phase-2 ("post-processing") candidate verifier for
 * KV-match.  This is the drop-in boundary: the four inline phase-2 blocks of the reference's
 * query engines are replaced by one call each.  A JNI shim (kv-match_amd/jni/) or any other FFI
 * binds exactly these symbols; INTEGRATION.md shows the reference-side binding.
 *
 * Reference paths are relative to src/main/java/cn/edu/fudan/dsm/kvmatch/ (abbreviated K/).
 *
 * Conventions shared with the reference:
 *   - series offsets are 1-based int32 (K/operator/file/TimeSeriesFileOperator.java:54-57);
 *   - all values are IEEE-754 binary64;
 *   - `left[i], right[i]` are the `validPositions` intervals handed over by phase 1, sorted,
 *     disjoint and non-adjacent (result of sortAndMergeIntervals, K/QueryEngine.java:664-693);
 *     only getLeft()/getRight() are read by phase 2;
 *   - `shift` = (lastSegment - 1) * WuList[0]  (K/QueryEngine.java:346; WuList[0] = 25);
 *   - candidate window start offsets of interval i are
 *         max(1, left[i]-shift) .. min(right[i]-shift, n-m+1)      (K/QueryEngine.java:346-354);
 *   - cntCandidate = sum(right[i]-left[i]+1), counted before clipping (K/QueryEngine.java:344);
 *   - answers are (offset, sqrt(distance^2)) sorted by distance, ties in ascending offset
 *     (stable sort of an offset-ascending list, K/QueryEngine.java:373).
 *
 * Threading: one kvm_ctx per engine instance; calls on one ctx must be serialised by the caller
 * (the reference engines are not re-entrant either).  A ctx owns one HIP stream on one device, its
 * pinned staging block and its scratch buffers.  Different contexts may be used from different host
 * threads at the same time -- several queries in flight against one resident series: kvm_share_series.
 * Multi-GPU = a kvm_group (below): one call per query over all the GPUs of a node, in one process or one process per GPU.
 * Process-wide state: (i) one pinned host block that the index-build calls (kvm_index_rows_build*) reuse
 * across calls and contexts, behind a mutex (calls that need it at the same time allocate their own);
 * (ii) a thread_local memo of the ED pruner's launch-shape model.  Nothing else is shared.
 *
 * Environment (read ONCE per process, at the first call that asks; all optional, none changes an answer):
 *   KVM_TRACE=1          per-stage host timings of every call on stderr
 *   KVM_DEBUG_SYNC=1     DTW calls wait for the stream after every stage and name it on stderr
 *   KVM_SERIALIZE_ENQUEUE=1  the default of KVM_OPT_SERIALIZE_ENQUEUE for contexts opened by this process
 *   KVM_LIB=<path>       (Python binding only) load another build of this library
 * Nothing else is read from the environment: the launch-shape experiments of rounds 2 - 5 are KVM_OPT_ED_FORCE (a per-context
 * option the tests use to drive every form of the pruner), and the library does not look for a profiler -- a host that runs
 * under one sets KVM_OPT_SERIALIZE_ENQUEUE itself (bench.py does, and says so in its line).
 *
 * DTW answer semantics.  The reference's DTW engines do not report {candidates with banded DTW <= eps^2}:
 * DtwUtils.dtw returns `min_cost + cb[i+r+1]` as soon as that lower bound is >= eps^2 (K/utils/DtwUtils.java:
 * 324-326), the LB_Keogh loops stop at the first partial sum >= eps^2 (:210, :242) and every caller accepts
 * `value <= eps^2` -- a bound that lands EXACTLY on eps^2 makes the candidate an answer at distance eps.  On
 * continuous data such ties have measure zero; on integer-valued data with an integer eps^2, and for eps = 0,
 * they are common.  kvm_verify_rsm_dtw / kvm_verify_cnsm_dtw reproduce them: same answers, same distances as the
 * Java loops (KVM_STAT_DTW_SEQUENTIAL counts the candidates that needed the sequential tie path).  The order cNSM-DTW
 * sums its LB_Keogh bounds in is the permutation java.util.TimSort (JDK 8) leaves under NormQueryEngineDtw's
 * contract-breaking comparator (K/NormQueryEngineDtw.java:475-478), restated in the library (kvm_cnsm_dtw_order,
 * KVM_OPT_SORT_VIOLATION) -- including the queries on which that sort throws.
 *
 * Every function returns KVM_OK (0) or a negative status; kvm_last_error() gives the message.
 * The library never falls back to a CPU path: without a usable HIP device kvm_open fails.
 */
#ifndef KVMATCH_H
#define KVMATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVM_OK 0
#define KVM_ERR_INVALID_HANDLE (-1)
/* maps to java.lang.IllegalArgumentException (K/operator/file/TimeSeriesFileOperator.java:55-57;
 * parameter validation of K/NormQueryEngineDtw.java:112-162) */
#define KVM_ERR_ILLEGAL_ARGUMENT (-2)
/* maps to java.io.IOException (propagates out of query(), K/QueryEngine.java:155) */
#define KVM_ERR_IO (-3)
#define KVM_ERR_DEVICE (-5)        /* HIP runtime failure (also -> IOException in the shim) */
#define KVM_ERR_NO_SERIES (-6)     /* verify called before a series was loaded */
#define KVM_ERR_UNSUPPORTED (-7)   /* parameter outside what the kernels are built for */
#define KVM_ERR_INDEX_OUT_OF_BOUNDS (-8) /* the reference ends in ArrayIndexOutOfBoundsException on this input
                                            (KVM_OPT_STRICT_BOUNDS; the shim throws the same) */

#define KVM_ERR_NO_MEMORY (-9)     /* host allocation failed inside the library (std::bad_alloc): no C++ exception ever crosses
                                      this boundary; -> java.lang.OutOfMemoryError in the shim */

typedef struct kvm_ctx kvm_ctx;
typedef struct kvm_result kvm_result;

/* ---- lifecycle ------------------------------------------------------------------------------ */

/* Replaces nothing in the reference (there is no device layer); the analogue is the engine
 * constructor QueryEngine(int n, String storageType) (K/QueryEngine.java:70).  `device` is the HIP
 * device ordinal as seen by this process. */
int kvm_open(int device, kvm_ctx **ctx);
int kvm_close(kvm_ctx *ctx);
const char *kvm_last_error(const kvm_ctx *ctx); /* ctx may be NULL: returns the last open() error */
const char *kvm_version(void);

/* ---- series staging: TimeSeriesOperator (K/operator/TimeSeriesOperator.java:29-54) ------------- */

/* Stage a native-endian fp64 host array of n values into HBM (once) and build the
 * block-restarted prefix sums of x and x^2 used for per-window mean/std.  Replaces the
 * per-interval readTimeSeries(begin, len) of K/QueryEngine.java:352 (cf. the "whole series in
 * memory" precedent K/operator/memory/TimeSeriesMemoryOperator.java:41-66). */
int kvm_load_series_f64(kvm_ctx *ctx, const double *host_values, int64_t n);

/* Same, from the reference's on-disk/wire encoding: n big-endian fp64 values
 * (K/common/entity/TimeSeriesNode.java:38-52, K/DataGenerator.java:111).  The byte swap runs on the GPU. */
int kvm_load_series_be64(kvm_ctx *ctx, const void *be_bytes, int64_t n);

/* Same, reading files/data-{N} (K/operator/file/TimeSeriesFileOperator.java:50-52). */
int kvm_load_series_file(kvm_ctx *ctx, const char *path, int64_t n);

/* Stage from an fp64 array already resident in this device's memory (e.g. a torch tensor's
 * data_ptr()): one device-to-device copy into the ctx's padded buffer; the caller's array is not
 * referenced after the call returns. */
int kvm_attach_series_device(kvm_ctx *ctx, const double *device_values, int64_t n);

/* Use the series `owner` has staged (values and prefix sums, read-only after staging) without copying it: one more
 * context -- its own stream, staging and scratch buffers -- on the same device, so that several queries can be in flight
 * against one resident series, each from its own host thread (a context is not re-entrant, like the reference's engine
 * instances, K/QueryEngine.java:61-68; several engines over one data file are the reference's analogue).  `owner` must
 * stay open, and must not load another series, for as long as `ctx` uses it.  Data envelopes (DTW) are cached per
 * context. */
int kvm_share_series(kvm_ctx *ctx, const kvm_ctx *owner);

int64_t kvm_series_length(const kvm_ctx *ctx);

/* Upper / lower envelope of radius rho of the staged series at points first .. first + count - 1 (1-based), into host
 * arrays: DtwUtils.lowerUpperLemire over the whole series as one chunk (K/utils/DtwUtils.java:93-134; a pure selection,
 * max / min of data[max(1, p - rho) .. min(n, p + rho)]).  Built on the GPU once per rho and cached; the DTW engines use
 * it for every point at least rho away from both ends of the chunk the reference would have fetched (the chunk-end
 * values are formed per candidate). */
int kvm_data_envelope(kvm_ctx *ctx, int32_t rho, int32_t first, int64_t count, double *upper, double *lower);

/* TimeSeriesOperator.readTimeSeries(left, length) served from HBM
 * (K/operator/TimeSeriesOperator.java:38; same range check and error as
 * K/operator/file/TimeSeriesFileOperator.java:55-57).  `out` receives `length` host doubles. */
int kvm_read_time_series(kvm_ctx *ctx, int32_t left, int32_t length, double *out);

/* Sliding window means for the index build: out[i] = mean(x[i .. i+w-1]) for every 1-based window
 * start i+1 in [first, first+count), computed from the block prefix sums (O(1) per window).  Replaces the running-sum
 * loop of K/IndexBuilder.java:232-247 (`ex += d; mean = ex / w; ... ex -= t[j]`, restarted every EPOCH = 100000 points
 * "for reducing the floating point error"); values agree with it to ~1e-14 relative, not bit for bit (the index keys
 * are these means rounded to 0.05, K/utils/MeanIntervalUtils.java:51-61).  `out` is a HOST array of `count` doubles. */
int kvm_window_means(kvm_ctx *ctx, int32_t w, int32_t first, int64_t count, double *out);

/* Index build, step 1 ("fixed-width index rows") of K/IndexBuilder.java:186-300 for one window length w, bit for bit:
 * the row key MeanIntervalUtils.toRound(mean) (K/utils/MeanIntervalUtils.java:50-60) of every window position, with
 * `mean = ex / w` from the reference's own running sums (restarted every EPOCH = 100000 points, replayed exactly on the
 * GPU), and consecutive positions with the same key joined into intervals of at most IndexNode.MAXIMUM_DIFF - 1 = 255
 * positions (:267).  The result lists the intervals in position order: keys[i], first[i], last[i] (1-based window
 * positions).  Grouping them by key into IndexNode rows (:270-284) and step 2 (:308-347) stay on the host.
 * 1 <= w <= min(n, 100000).  The result is owned by the library until kvm_index_rows_free. */
typedef struct kvm_index_rows kvm_index_rows;
int kvm_index_rows_build(kvm_ctx *ctx, int32_t w, kvm_index_rows **out);
int64_t kvm_index_rows_count(const kvm_index_rows *r);
const double *kvm_index_rows_keys(const kvm_index_rows *r);
const int32_t *kvm_index_rows_first(const kvm_index_rows *r);
const int32_t *kvm_index_rows_last(const kvm_index_rows *r);
void kvm_index_rows_free(kvm_index_rows *r);
/* The same intervals collected into the rows the reference builds -- one IndexNode per distinct key
 * (K/IndexBuilder.java:270-284) -- and ordered the way step 2 walks them, by DESCENDING key (Double.compareTo; :318):
 * keys/first/last are sorted by (key descending, position ascending), group g is the row with key
 * group_keys[g] and holds the intervals [group_offsets[g], group_offsets[g + 1]).  kvm_index_rows_groups is 0 for a
 * result of kvm_index_rows_build. */
int kvm_index_rows_build_grouped(kvm_ctx *ctx, int32_t w, kvm_index_rows **out);
int64_t kvm_index_rows_groups(const kvm_index_rows *r);
const double *kvm_index_rows_group_keys(const kvm_index_rows *r);
const int32_t *kvm_index_rows_group_offsets(const kvm_index_rows *r);

/* The whole IndexBuilder.run for one window length w (= Wu; K/IndexBuilder.java:186-347) and the bytes
 * IndexFileOperator.writeAll puts into files/index-{N}-{Wu} (K/operator/file/IndexFileOperator.java:127-165): step 1 on
 * the GPU (kvm_index_rows_build_grouped), then on the host step 2 -- adjacent rows merged by the reference's rule
 * (:308-347, K/utils/IndexNodeUtils.java:30-90) --, IndexNode.toBytesCompact (K/common/entity/IndexNode.java:61-100) per
 * row, the statistic triples (K/utils/ByteUtils.java:86-104) and the offset table.  Byte for byte the file the Java
 * builder writes for the same series.  KVM_ERR_ILLEGAL_ARGUMENT when the series holds no window of length w,
 * KVM_ERR_UNSUPPORTED when the file would reach 2 GiB (its offsets are Java ints).  The image is owned by the library
 * until kvm_index_file_free; kvm_index_file_write stores it at `path` (the reference's name is
 * "files/index-" + N + "-" + Wu).
 * Host side of a call: up to 16 threads of the library's own for its duration (the sparse rows at both ends of the key
 * range by teams of four working as a pipeline, placed on CPUs that share a last-level cache -- the calling thread's CPU
 * mask is changed for the time it works in such a team and put back before the call returns); between calls the process
 * keeps three buffers for the next build (pinned staging of step 1's rows, 8 B per interval; the finished rows, 8 B per
 * interval; the image of the last freed file): ~ 1 GB at n = 1e8, Wu = 25. */
typedef struct kvm_index_file kvm_index_file;
int kvm_index_build(kvm_ctx *ctx, int32_t w, kvm_index_file **out);
int64_t kvm_index_file_size(const kvm_index_file *f);
const uint8_t *kvm_index_file_bytes(const kvm_index_file *f);
#define KVM_INDEX_ROWS_BEFORE 0      /* rows (distinct keys) after step 1 */
#define KVM_INDEX_ROWS_AFTER 1       /* rows after step 2 = lines of the file */
#define KVM_INDEX_INTERVALS_BEFORE 2
#define KVM_INDEX_INTERVALS_AFTER 3
#define KVM_INDEX_STEP1_US 4         /* wall time of the three stages of the call, microseconds */
#define KVM_INDEX_STEP2_US 5
#define KVM_INDEX_ENCODE_US 6
int64_t kvm_index_file_info(const kvm_index_file *f, int32_t what);
int kvm_index_file_write(const kvm_index_file *f, const char *path);
void kvm_index_file_free(kvm_index_file *f);

/* Phase-2 time model for the phase-1 early-termination test (K/QueryEngine.java:316-327 and the same lines of the other
 * three engines): estimated T_2 in milliseconds of this library for `disjoint_windows` candidate intervals holding
 * `candidate_offsets` offsets and a query of `length` points,
 *     T_2 = C + A * disjoint_windows + (B * length + E) * (candidate_offsets / 1e5) + D * (length | length^2 / 1e3),
 * i.e. the reference's own formula (PHASE_2_TIME_ESTIMATE_COEFFICIENT_A/B/INTERCEPT, K/QueryEngine.java:55-57) plus the
 * terms a GPU call adds: C, its launch chain; D, the one unit of work that holds an answer and walks the whole window
 * (`length` points for the ED engines, one banded DP ~ length^2 for the DTW engines); E, the part of a candidate's cost
 * that does not grow with the query.  Constants fitted on an MI355X by
 * profiles/fit_cost_model.py -> profiles/r04_phase2_cost_model.json and compiled in from kv-match_amd/csrc/kvm_cost_model.inc
 * (the Java constants are A = 4.07 / B = 0.27 for RSM-ED and 9.72 / 0.0107 for the others: five to six orders of magnitude
 * larger); tests/test_zz_perf_gpu.py::test_phase2_time_model_against_measured_calls checks the estimate against
 * measured calls on held-out shapes (it fails beyond twice the worst residual of the variant's own fit).
 * `variant`: 0 rsm_ed, 1 cnsm_ed, 2 rsm_dtw, 3 cnsm_dtw.  Pure function; < 0 for an unknown variant. */
double kvm_phase2_time_estimate_ms(int32_t variant, int64_t disjoint_windows, int64_t candidate_offsets, int32_t length);
/* The same with the number of candidates that reach the banded DP (KVM_STAT_PASS_LB_KEOGH2 of an earlier call on the same
 * list, or a caller's own bound): the DTW engines' time follows it -- 3 or 700 DPs for the same number of offsets -- and
 * no a-priori model can know it.  dp_count < 0 or an ED variant: identical to kvm_phase2_time_estimate_ms.  num_cu: compute
 * units of the device (0 = 256, one MI355X). */
double kvm_phase2_time_estimate_dp_ms(int32_t variant, int64_t disjoint_windows, int64_t candidate_offsets, int32_t length,
                                      int64_t dp_count, int32_t num_cu);
/* The early-termination test itself, deterministic (K/QueryEngine.java:316-327: `if (i >= 5 && totalTimeUsageEstimated >
 * lastTotalTimeUsageEstimated) break; lastTotalTimeUsageEstimated = totalTimeUsageEstimated;`).  The reference adds the
 * WALL-CLOCK time of phase 1 so far to its phase-2 estimate, so `lastSegment` -- and with it the candidate list -- differs
 * from run to run; here the caller passes its phase-1 cost so far from a model of its own (`phase1_ms_so_far`: index rows
 * scanned times a constant, say), the phase-2 term is kvm_phase2_time_estimate_ms.  `segment_index` is the loop's `i`,
 * `*last_total_ms` the carried `lastTotalTimeUsageEstimated` (the reference starts it at Double.MAX_VALUE, :187; any start
 * value works: nothing stops before i = 5 and every earlier call overwrites it).  Returns 1 = stop (the
 * Java `break`), 0 = go on (and `*last_total_ms` is updated), < 0 for bad arguments. */
int32_t kvm_phase1_should_stop(int32_t variant, int32_t segment_index, double phase1_ms_so_far, int64_t disjoint_windows,
                               int64_t candidate_offsets, int32_t length, double *last_total_ms);

/* ---- phase 2: the four verification entry points ------------------------------------------------ */

/* The interval list.  left[i] / right[i] are `validPositions` as the engines hold them right before phase 2: the output
 * of sortAndMergeIntervals (sorted by left, disjoint, non-adjacent; kvm_sort_and_merge_intervals produces that form).
 * Lists with adjacent, overlapping, nested, repeated or empty (right < left) intervals are processed the way the Java
 * loops process them -- every interval scanned on its own, an offset covered twice answered twice, cntCandidate summing
 * right - left + 1 as it stands.  A list NOT sorted by left: the RSM engines treat intervals independently and return
 * the same (offset, distance) pairs as for the sorted list; the cNSM engines index their merged chunks with
 * `left - begin`, which is an exception in Java, and return KVM_ERR_ILLEGAL_ARGUMENT here. */

/* RSM-ED.  Replaces K/QueryEngine.java:338-363 (+ sort :373). */
int kvm_verify_rsm_ed(kvm_ctx *ctx, const double *query, int32_t m, const int32_t *left, const int32_t *right,
                      int32_t k, int32_t shift, double epsilon, kvm_result **result);

/* cNSM-ED.  Replaces K/NormQueryEngine.java:432-528; meanQ/stdQ are recomputed from `query` exactly
 * as K/NormQueryEngine.java:192-198 does. */
int kvm_verify_cnsm_ed(kvm_ctx *ctx, const double *query, int32_t m, const int32_t *left, const int32_t *right,
                       int32_t k, int32_t shift, double epsilon, double alpha, double beta, kvm_result **result);

/* RSM-DTW.  Replaces K/QueryEngineDtw.java:349-452.  `rho` is the resolved Sakoe-Chiba half-width
 * (see kvm_resolve_rho).  rho >= m, and an interval whose fetched data is no longer than rho points (one that starts beyond
 * offset n - m + 1), are inputs on which the reference ends in ArrayIndexOutOfBoundsException (DtwUtils.lowerUpperLemire
 * stores to u[i - r - 1] with a negative index): both DTW entry points return KVM_ERR_INDEX_OUT_OF_BOUNDS there, first
 * offender in the reference's order (KVM_OPT_STRICT_BOUNDS, default on; 0 computes the envelope's natural extension). */
int kvm_verify_rsm_dtw(kvm_ctx *ctx, const double *query, int32_t m, const int32_t *left, const int32_t *right,
                       int32_t k, int32_t shift, double epsilon, int32_t rho, kvm_result **result);

/* cNSM-DTW.  Replaces K/NormQueryEngineDtw.java:457-603. */
int kvm_verify_cnsm_dtw(kvm_ctx *ctx, const double *query, int32_t m, const int32_t *left, const int32_t *right,
                        int32_t k, int32_t shift, double epsilon, int32_t rho, double alpha, double beta,
                        kvm_result **result);

/* ---- several queries in flight (host layer; replaces nothing in the reference: its engines are single-threaded and not
 * re-entrant, K/QueryEngine.java:61-68 -- a server runs one engine instance per thread over the same data file) ---------- */

/* One phase-2 call as data: `variant` 0 rsm_ed, 1 cnsm_ed, 2 rsm_dtw, 3 cnsm_dtw; the other fields are the arguments of the
 * matching kvm_verify_* (rho / alpha / beta ignored where the variant has none).  All pointers are borrowed until
 * kvm_run_jobs returns. */
typedef struct kvm_job {
    int32_t variant, m;
    const double *query;
    const int32_t *left, *right;
    int32_t k, shift;
    double epsilon;
    int32_t rho, reserved;
    double alpha, beta;
} kvm_job;

/* Runs jobs[0..n_jobs) on the n_ctx contexts (normally one that owns a series and n_ctx - 1 that share it:
 * kvm_share_series), one host thread per context, every thread taking the next job in order as soon as its context is
 * free -- exactly what n_ctx threads calling kvm_verify_* in a loop do, without a foreign-language driver between the
 * calls.  results[i] / status[i] receive job i's kvm_result (caller frees) and status; after a failing job no further
 * job is started.  Returns KVM_OK or the status of the first (lowest-index) job that failed; that job's message is then
 * what kvm_last_error(ctxs[0]) returns, whichever context ran it. */
int kvm_run_jobs(kvm_ctx *const *ctxs, int32_t n_ctx, const kvm_job *jobs, int64_t n_jobs, kvm_result **results,
                 int32_t *status);

/* The same loop for the device-resident variant (the multi-GPU host layer: every rank keeps several sharded queries in
 * flight and ONE thread of the caller issues the collectives in query order).  Slot t = a context with its own stream and
 * output buffers; job i runs on slot i mod n_slots: kvm_verify_device + kvm_pack_answers_device, enqueued, not waited for.
 * Hand-shake through two int32 arrays of n_jobs entries owned by the caller: the library sets produced[i] = 1 once job i's
 * kernels are enqueued on its slot's stream (a job that failed: its status in status[i], and {-1, 0} stored into the slot's
 * packed block on that stream, so that a collective over the blocks tells every rank) and does not start job i + n_slots
 * before the caller has set consumed[i] (the slot's buffers are free again); *stop != 0 ends the stream early.  Blocks until
 * every job is enqueued or the stream is stopped: call it from a thread of its own.  cnt_candidate[i] as kvm_verify_device. */
typedef struct kvm_device_slot {
    kvm_ctx *ctx;
    void *stream;        /* hipStream_t of the slot */
    int32_t *d_off;      /* device, cap entries */
    double *d_dist;      /* device, cap entries */
    int64_t cap;
    int64_t *d_count;    /* device */
    double *d_packed;    /* device, 2 * (pack_cap + 1) doubles */
    int64_t pack_cap;
    void *ready;         /* hipEvent_t or NULL: recorded on the slot's stream behind each job's last kernel (or its failure
                            mark), BEFORE produced[i] is set -- what the caller's exchange stream waits for.  (Added in
                            version 0.5, at the end: a caller built against the shorter struct must be rebuilt) */
} kvm_device_slot;
int kvm_run_device_jobs(const kvm_device_slot *slots, int32_t n_slots, const kvm_job *jobs, int64_t n_jobs,
                        volatile int32_t *produced, const volatile int32_t *consumed, const volatile int32_t *stop,
                        int64_t *cnt_candidate, int32_t *status);

/* `r <= 1 -> floor(r*m) else floor(r)`  (K/QueryEngineDtw.java:135-140). */
int32_t kvm_resolve_rho(double r, int32_t m);

/* sortAndMergeIntervals (K/QueryEngine.java:664-693; the same method in the other three engines): what every engine
 * applies to phase 1's `validPositions` right before phase 2 -- sort by `left` (stable), then join every interval whose
 * left - 1 <= the running end.  The verify entry points expect their list in this form (sorted, disjoint, non-adjacent).
 * Host only (no GPU call).  In: k intervals in any order; out_left/out_right: room for k entries (may alias the inputs
 * only if they are the same arrays as left/right).  Returns the number of merged intervals (k <= 1: the list as it is),
 * or KVM_ERR_ILLEGAL_ARGUMENT.  Interval.epsilon (min over the joined intervals) is not read by phase 2 and not carried. */
int64_t kvm_sort_and_merge_intervals(const int32_t *left, const int32_t *right, int64_t k, int32_t *out_left,
                                     int32_t *out_right);

/* The permutation cNSM-DTW sums its LB_Keogh bounds in: order[i] = index of the i-th element of tmpQ after
 * `Arrays.sort(tmpQ, (o1, o2) -> (int) (Math.abs(o2.value) - Math.abs(o1.value)))` on the z-normalised query
 * (K/NormQueryEngineDtw.java:470-484), as JDK 8's TimSort leaves it.  Host only (no GPU call); `query` is the raw
 * query, z-normalised here like the engine does.  Returns KVM_OK, or KVM_ERR_ILLEGAL_ARGUMENT for a query on which
 * that sort throws "Comparison method violates its general contract!" (see KVM_OPT_SORT_VIOLATION): a caller can
 * ask before submitting.  `order` may be NULL when only the verdict is wanted. */
int kvm_cnsm_dtw_order(const double *query, int32_t m, int32_t *order);

/* ---- results: List<Pair<Integer, Double>> answers + statistics[2..4] ---------------------------- */

int64_t kvm_result_count(const kvm_result *r);        /* answers.size()   -> statistics[4] (K/QueryEngine.java:370) */
int64_t kvm_result_candidates(const kvm_result *r);   /* cntCandidate     -> statistics[3] (K/QueryEngine.java:369) */
double kvm_result_elapsed_ms(const kvm_result *r);    /* T_2 wall ms      -> statistics[2] (K/QueryEngine.java:368) */
const int32_t *kvm_result_offsets(const kvm_result *r);   /* Pair.getFirst(), sorted by distance */
const double *kvm_result_distances(const kvm_result *r);  /* Pair.getSecond() */
/* Work counters (no reference counterpart; for measurement): see KVM_STAT_* */
int64_t kvm_result_stat(const kvm_result *r, int32_t which);
void kvm_result_free(kvm_result *r);

#define KVM_STAT_WINDOWS 0       /* candidate windows actually evaluated (after clipping) */
#define KVM_STAT_PASS_FILTER 1   /* passed the (alpha, beta) filter (cNSM) */
#define KVM_STAT_PASS_LB_KIM 2   /* not dropped by the LB_Kim screen (DTW) */
#define KVM_STAT_PASS_LB_KEOGH 3 /* not dropped by the LB_Keogh screen on the query envelope (DTW) */
#define KVM_STAT_PASS_LB_KEOGH2 4 /* passed the whole cascade in the exact stage = DP runs started (DTW) */
#define KVM_STAT_KERNEL_NS 5     /* device time of the dominant kernel(s), ns (HIP events) */
#define KVM_STAT_RECHECKED 6     /* knife-edge candidates re-evaluated in reference order */
#define KVM_STAT_DTW_SEQUENTIAL 7 /* DTW: candidates within rounding of a tie with eps^2, decided by the sequential
                                   * reference-ordered path (0 on continuous data) */
#define KVM_STAT_COUNT 8

/* ---- knobs (measurement only; defaults reproduce the reference's behaviour) ---------------------- */

#define KVM_OPT_EARLY_ABANDON 1  /* 1 (default): stop a distance once it exceeds eps^2, as the
                                    reference does; 0: evaluate every candidate over all m points */
#define KVM_OPT_LB_CASCADE 2     /* 1 (default): LB_Kim / LB_Keogh screens in front of the exact DTW stage; 0: every
                                    filter-passing candidate goes to the exact stage.  Same answers either way. */
#define KVM_OPT_ED_KERNEL 3      /* ED pruning kernel: 0 auto (default = 3), 1 wave-per-candidate (the cross-check of
                                    the tests), 3 LDS-tiled correlation form.  Identical answers.  (2 and 4 -- the
                                    direct LDS-tiled and the lane-granular pruners of rounds 2 - 3 -- were removed and
                                    are refused.) */
#define KVM_OPT_TIMING 4         /* 1: record HIP-event kernel times into KVM_STAT_KERNEL_NS */
#define KVM_OPT_DP_KERNEL 5      /* banded-DP form: 0 auto (rho <= 511: band in registers; the packed form -- any number of
                                    cells per lane -- for rho = 128 .. 191 and 256 .. 447), 1 generic LDS form, 2 one candidate per
                                    wave, 3 packed form wherever it exists (also two candidates per wave at the other rho in
                                    64 .. 255, which the default never picks).  Identical answers. */
#define KVM_OPT_SORT_VIOLATION 6 /* cNSM-DTW only.  NormQueryEngineDtw sorts its query with a comparator that breaks
                                    the Comparator contract (K/NormQueryEngineDtw.java:475-478); for some queries
                                    (about half of white-noise ones, a few percent of random walks, no smooth one)
                                    java.util.TimSort then throws IllegalArgumentException("Comparison method
                                    violates its general contract!") and the reference's query() dies with it.
                                    0 (default): kvm_verify_cnsm_dtw returns KVM_ERR_ILLEGAL_ARGUMENT with that
                                    message for exactly those queries; 1: answer them anyway (lower bounds summed in
                                    index order -- the answers are those of any successful order except at exact
                                    ties with eps^2). */
#define KVM_OPT_UCR_LOOPS 7      /* 0 (default): the engines' loops.  1: the loops of the reference's index-free executors
                                    where they differ (set only by their mirrors, kvmatch.UcrEdQueryExecutor /
                                    UcrDtwQueryExecutor; the candidate list must then be the executor's scan -- consecutive
                                    intervals in offset order).  kvm_verify_cnsm_ed: UcrEdQueryExecutor.distance() runs
                                    `for (...; i < m && sum < bsf; ...)` (experiments/ucr/UcrEdQueryExecutor.java:91-98), so a
                                    partial sum exactly on eps^2 ends it and the window is an answer at distance sqrt(eps^2);
                                    with eps^2 == 0 every window passing (Alpha, Beta) is an answer at distance 0.
                                    kvm_verify_cnsm_dtw: UcrDtwQueryExecutor never clears cb1 / cb2 between windows
                                    (experiments/ucr/UcrDtwQueryExecutor.java:157-161): a window whose LB_Keogh loop stops
                                    exactly on eps^2 runs its DP with what earlier windows left in the rest of the array. */
#define KVM_OPT_STRICT_BOUNDS 8  /* 1 (default): the DTW entry points fail where the reference dies in
                                    DtwUtils.lowerUpperLemire (K/utils/DtwUtils.java:81-90: `u[i - r - 1]` with an array no
                                    longer than the band): rho >= m (the query envelope), or an interval whose fetched data
                                    holds <= rho points (a stub beyond the last window offset) -> KVM_ERR_INDEX_OUT_OF_BOUNDS,
                                    which the JNI shim throws as ArrayIndexOutOfBoundsException.  0: compute the envelope's
                                    natural extension (window clipped at the array's ends) and answer. */
#define KVM_OPT_IN_FLIGHT 9      /* 1 (default) .. 64: how many contexts the caller keeps busy on this GPU at the same time with
                                    threads of its own (kvm_run_jobs / kvm_run_device_jobs know theirs and need no option).  A
                                    hint to the ED pruner's launcher, results are the same: with other launches next to its
                                    own it plans a list for throughput (no window split beyond its share of the chip, the
                                    short tiles in their denser form) instead of for the shortest single launch.  Replaces
                                    nothing in the reference. */
#define KVM_OPT_ED_FORCE 10      /* 0 (default): the ED pruner's launcher chooses its launch shape (tile classes, candidates per
                                    lane, window split) from its model.  Otherwise a bit field that FORCES parts of the shape --
                                    for the tests, which drive every form over the same inputs, and for measurements; the
                                    answers are the same: bits 0-3 window split of class-A units (1, 2 or 4; 0 = model),
                                    bits 4-7 of class-B / C units, bits 8-9 class A on 16 candidates per lane (1 = never,
                                    2 = wherever the form exists), bits 10-11 class B on the four-lane rows (1 never, 2 with
                                    16 per lane in class A), bits 12-13 tile classes of host-planned lists (1 = A / B, 2 = A /
                                    B / C), bit 14 device-built (cNSM) lists of any size through the 16-per-lane form, bit 15
                                    device-built lists never split. */
#define KVM_ED_FORCE(split_a, split_b, a16, b16, classes, dev_a16_any, dev_no_split) \
    ((int64_t)(split_a) | ((int64_t)(split_b) << 4) | ((int64_t)(a16) << 8) | ((int64_t)(b16) << 10) | ((int64_t)(classes) << 12) | \
     ((int64_t)(dev_a16_any) << 14) | ((int64_t)(dev_no_split) << 15))
#define KVM_OPT_ED_LAST_SHAPE 12  /* read only (kvm_get_option): the launch shape of this context's most recent LDS-tiled ED pruner
                                    launch, in the layout of KVM_OPT_ED_FORCE -- bits 0-3 / 4-7 the window splits, bits 8-9 class A on
                                    8 (1) or 16 (2) candidates per lane, bits 10-11 class B on 4 (1) or 8 (2) -- plus bits 16-23 the
                                    queries in flight the launcher planned for.  0: no such launch yet. */
#define KVM_OPT_SERIALIZE_ENQUEUE 11 /* 0 (default, or KVM_SERIALIZE_ENQUEUE=1 in the environment when the context is opened): host
                                    threads enqueue freely.  1: the ENQUEUE phases of this context's verification calls (plan,
                                    copies, launches; not the waits) hold one process-wide lock -- the kernels of different
                                    contexts still overlap on the GPU, the host threads take turns issuing them.  For hosts that
                                    run under a tool that hooks kernel dispatches: rocprofv3's kernel tracing falls over when
                                    several threads enqueue at the same moment (profiles/r05_prof_native_crash.log).  Costs
                                    nothing where every point is evaluated and a third of an early-abandoning stream's rate
                                    (39 -> 59 us per query, profiles/r06_enqueue_lock_cost.log), so it is not the default. */
int kvm_set_option(kvm_ctx *ctx, int32_t option, int64_t value);
int64_t kvm_get_option(const kvm_ctx *ctx, int32_t option);
/* Device time (ns, HIP events on the launch stream) of the verification kernels of the most recent
 * kvm_verify_* / kvm_verify_device call on this ctx; 0 unless KVM_OPT_TIMING is set. */
int64_t kvm_last_kernel_ns(const kvm_ctx *ctx);

/* What the device under this run sustains, for the roofline lines of a benchmark (vendor peaks are quoted, these are
 * measured): `what` 0 = HBM read stream, 16 B per lane over 1 GiB, GB/s (what the ED pruner does to HBM is read windows); 1 = fp64
 * FMA issue rate, register-only, TFLOP/s (FMA = 2); 2 = fp64 issue rate of the banded DP's cell mix (sub, mul, min, min, add; no
 * FMA), Tops/s.  Allocates and frees its own 1 GiB (what = 0); waits for the context's stream.  Replaces nothing in the reference. */
int kvm_measure_device(kvm_ctx *ctx, int32_t what, double *value);

/* ---- device-resident variant (multi-GPU host plumbing: no host copies, caller's stream) ----------- */

/* Same computation as kvm_verify_*, but survivors stay in device memory so that the host layer can
 * hand them to one RCCL all-gather.  `variant`: 0 rsm_ed, 1 cnsm_ed, 2 rsm_dtw,
 * 3 cnsm_dtw.  d_out_off / d_out_dist are device buffers of `cap` entries, d_out_count a device
 * int64.  Unsorted.  `stream` is a hipStream_t (NULL = the ctx stream).  The ED variants return without
 * waiting for the stream; the DTW variants wait for it once (the number of candidates that need the sequential tie
 * path is only known on the device, and that path's scratch is sized on the host).  Ordering: the next call on this
 * context is ordered behind this one's kernels on the device (stream order, or an event wait when it runs on another
 * stream); the host only waits until the call's inputs have left the pinned staging buffer, so back-to-back calls overlap
 * their host work with the previous call's kernels.  The caller must not read d_out_* before `stream` has caught up. */
int kvm_verify_device(kvm_ctx *ctx, int32_t variant, const double *query, int32_t m, const int32_t *left,
                      const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho, double alpha,
                      double beta, int32_t *d_out_off, double *d_out_dist, int64_t cap, int64_t *d_out_count,
                      int64_t *cnt_candidate, void *stream);

/* Packs the device-resident answers of kvm_verify_device for a fixed-size exchange: d_packed (fp64, 2*(pack_cap+1)
 * entries) receives {count, 0} followed by the first min(count, pack_cap) (offset, distance) pairs.  One kernel on
 * `stream`; used by the multi-GPU host layer before its single all-gather. */
int kvm_pack_answers_device(kvm_ctx *ctx, const int32_t *d_off, const double *d_dist, const int64_t *d_count,
                            int64_t pack_cap, double *d_packed, void *stream);

/* ---- one engine over the GPUs of a node: candidate shards, ONE all-gather of the answers --------------- */

/* A reference engine is one object per data file (K/QueryEngine.java:70) making one phase-2 pass per query (:155, the loop
 * over the intervals :343-363).  A kvm_group is that object over several GPUs: the series replicated on every device, the
 * interval list of a call dealt out to the devices (kvm_split_intervals), every device verifying its shard with the answers
 * left in HBM (kvm_verify_device + kvm_pack_answers_device, one native host thread per device and slot), ONE ncclAllGather
 * (RCCL over xGMI, issued from native code: librccl is opened by the library itself, no Python and no torch on the path) of
 * a fixed-size block [count | first pack_cap (offset, distance) pairs] per rank, then merge and the (distance, offset) sort
 * of K/QueryEngine.java:373 on the host -- the same kvm_result a single-GPU call returns, bit for bit.  Only when some
 * rank holds more than pack_cap answers does a second, exact-size exchange follow.  A shard that fails takes part in the
 * exchange with a count of -1: every rank learns it there and returns an error for that call; nobody is left waiting.
 *
 * Two ways to form a group:
 *   kvm_open_devices  ONE process drives n_dev GPUs (a JVM with 8 devices): ncclCommInitAll inside the process.
 *   kvm_open_rank     one process per GPU (the launchers of this repository's bench.py, MPI-style hosts): rank `rank` of
 *                     `world`; `id` = the KVM_GROUP_ID_BYTES bytes kvm_group_unique_id wrote on ONE rank, carried to the
 *                     others by the host's own means (a file, a socket, a launcher's store); every rank passes the same
 *                     interval lists to the same sequence of calls and ends with the full answer list.
 * `in_flight` (1 .. 64): slots per device = sharded queries kvm_run_sharded_jobs overlaps on every device (each slot is a
 * context of its own over the device's resident series: kvm_share_series).  `flags`: KVM_GROUP_RCCL_ALWAYS makes a group of
 * ONE rank exchange through RCCL as well (otherwise it copies its block to the host directly and never opens librccl).
 * Calls on one group must be serialised by the caller (as on a kvm_ctx). */
typedef struct kvm_group kvm_group;
#define KVM_GROUP_ID_BYTES 128
#define KVM_GROUP_RCCL_ALWAYS 1
#define KVM_ERR_SHARD_FAILED (-10) /* another rank failed on its shard of the call (its own status is what THAT rank returns);
                                      -> IOException in the shim */
int kvm_open_devices(const int32_t *devices, int32_t n_dev, int32_t in_flight, int32_t flags, kvm_group **group);
int kvm_group_unique_id(void *id, int32_t id_bytes);
int kvm_open_rank(int32_t device, int32_t rank, int32_t world, const void *id, int32_t id_bytes, int32_t in_flight,
                  int32_t flags, kvm_group **group);
int kvm_group_close(kvm_group *group);
const char *kvm_group_last_error(const kvm_group *group);  /* group may be NULL: the last kvm_open_devices / kvm_open_rank error */
int32_t kvm_group_world(const kvm_group *group);          /* ranks = shards of a call */
int32_t kvm_group_local_devices(const kvm_group *group);  /* devices this process drives */
int32_t kvm_group_rccl_ranks(const kvm_group *group);     /* ranks of the RCCL communicator; 0: none was made */
/* the context of slot `slot` of local device `local_device` (slot 0 owns the device's series): for kvm_get_option,
 * kvm_read_time_series, kvm_last_kernel_ns ...; NOT for verification calls while a group call runs */
kvm_ctx *kvm_group_ctx(kvm_group *group, int32_t local_device, int32_t slot);
/* Where the threads of the most recent kvm_run_sharded_jobs / kvm_verify_sharded spent their time, nanoseconds summed over
 * the jobs (measurement only).  The calling thread: waiting for a job's exchange to be issued, waiting for its block to
 * arrive, merge + sort.  The slot threads, summed over them: waiting for their slot to be handed back, splitting the list (or
 * waiting for the thread that does), inside kvm_verify_device + kvm_pack_answers_device, waiting for the turn to issue the
 * job's exchange, and the exchange itself -- preparing it, the all-gather (host time inside RCCL), the readback launch + event.
 * KVM_GROUP_STAT_JOBS: jobs of that call. */
#define KVM_GROUP_STAT_JOBS 0
#define KVM_GROUP_STAT_WAIT_ISSUED_NS 1
#define KVM_GROUP_STAT_PREPARE_NS 2
#define KVM_GROUP_STAT_ALL_GATHER_NS 3
#define KVM_GROUP_STAT_COPY_NS 4
#define KVM_GROUP_STAT_WAIT_BLOCK_NS 5
#define KVM_GROUP_STAT_MERGE_NS 6
#define KVM_GROUP_STAT_SLOT_WAIT_NS 7
#define KVM_GROUP_STAT_SLOT_SPLIT_NS 8
#define KVM_GROUP_STAT_SLOT_ENQUEUE_NS 9
#define KVM_GROUP_STAT_SLOT_TURN_NS 10
#define KVM_GROUP_STAT_COUNT 11
int64_t kvm_group_stat(const kvm_group *group, int32_t which);
#define KVM_GROUP_PACK_CAP 1    /* answers per rank the one fixed-size exchange carries (default 4096; >= 1) */
#define KVM_GROUP_SPLIT_MODE 2  /* KVM_SPLIT_* (default KVM_SPLIT_AUTO) */
int kvm_group_configure(kvm_group *group, int32_t what, int64_t value);
/* kvm_load_series_* on every local device ("replicates to every device"), then kvm_share_series for the slots */
int kvm_group_load_series_f64(kvm_group *group, const double *host_values, int64_t n);
int kvm_group_load_series_be64(kvm_group *group, const void *be_bytes, int64_t n);
int kvm_group_load_series_file(kvm_group *group, const char *path, int64_t n);
int kvm_group_set_option(kvm_group *group, int32_t option, int64_t value);  /* kvm_set_option on every context of the group */

/* The phase-2 pass of one query on every GPU of the group.  `variant` 0 rsm_ed, 1 cnsm_ed, 2 rsm_dtw, 3 cnsm_dtw; the
 * other arguments are those of the matching kvm_verify_* (rho / alpha / beta ignored where the variant has none) and
 * `left`/`right` is the WHOLE list (every rank of a kvm_open_rank group passes the same one).  Replaces the same blocks
 * as kvm_verify_* (K/QueryEngine.java:338-363 and its three siblings); the result is the one kvm_verify_* returns for the
 * unsharded list -- offsets, order, distances, cntCandidate -- with the work counters (kvm_result_stat) left at 0. */
int kvm_verify_sharded(kvm_group *group, int32_t variant, const double *query, int32_t m, const int32_t *left,
                       const int32_t *right, int32_t k, int32_t shift, double epsilon, int32_t rho, double alpha,
                       double beta, kvm_result **result);
/* A stream of such passes, `in_flight` of them overlapping on every device: the slots' threads enqueue, the calling thread
 * issues the all-gathers in job order and merges (the all-gather and host copy of job i overlap the kernels of job i+1).
 * results[i] / status[i] as kvm_run_jobs; after a failing job no further job is started and the call returns its status
 * (every rank of the group stops at the same job). */
int kvm_run_sharded_jobs(kvm_group *group, const kvm_job *jobs, int64_t n_jobs, kvm_result **results, int32_t *status);

/* The split rule, host only: shard `part` of `parts` of an interval list, balanced by candidate count; every candidate
 * offset lands in exactly one shard.
 *   KVM_SPLIT_INTERLEAVED  whole intervals are dealt out in rounds of `parts` consecutive intervals, the longest of a round to
 *       the shard that is lightest so far: neighbouring intervals land on different GPUs (the expensive candidates of a DTW
 *       query cluster around the matches), and because no interval is cut a sharded cNSM run restarts the reference's running
 *       sums (K/NormQueryEngine.java:487) exactly where the unsharded one does.
 *   KVM_SPLIT_CONTIGUOUS   contiguous slices of the candidate sequence, cut inside an interval where a boundary falls.
 *   KVM_SPLIT_AUTO         interleaved when the list holds at least 8 intervals per part, else contiguous.
 * Empty intervals (right < left) belong to no shard.  out_left / out_right: room for k entries, or both NULL to count only.
 * Returns the number of intervals of the shard, or KVM_ERR_ILLEGAL_ARGUMENT. */
#define KVM_SPLIT_AUTO 0
#define KVM_SPLIT_INTERLEAVED 1
#define KVM_SPLIT_CONTIGUOUS 2
int64_t kvm_split_intervals(const int32_t *left, const int32_t *right, int64_t k, int32_t parts, int32_t part, int32_t mode,
                            int32_t *out_left, int32_t *out_right);

/* Layout version of the structs that cross this boundary (kvm_job, kvm_device_slot): 6 since kvm_device_slot.ready (0.5)
 * and the group entry points.  A binding asserts it at load time instead of mis-striding an array of slots. */
#define KVM_ABI_VERSION 6
int32_t kvm_abi_version(void);

/* ---- synthetic inputs: seeded restatement of K/DataGenerator.java:80-119 ------------------------ */

/* Fills out[0..n) with the reference generator's distribution (random-walk / Gaussian / mixed-sine
 * segments), deterministically from `seed`.  Host-side; used by tests and bench.py. */
int kvm_gen_series(uint64_t seed, int64_t n, double *out);

/* Seeded phase-1-shaped interval list (sorted, disjoint, non-adjacent, geometric lengths, total
 * about `target_candidates`), always covering `must_cover[0..n_must)`.  Returns the number of
 * intervals written (<= cap) or a negative status. */
int64_t kvm_gen_intervals(uint64_t seed, int64_t n, int32_t m, int64_t target_candidates, double mean_len,
                          const int32_t *must_cover, int32_t n_must, int32_t *left, int32_t *right, int64_t cap);

/* Write / read files/data-{N} in the reference's format (flat big-endian fp64). */
int kvm_write_series_file(const char *path, const double *values, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* KVMATCH_H */
