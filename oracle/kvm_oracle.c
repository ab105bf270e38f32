/*
 * kvm_oracle.c -- CPU restatement of KV-match's phase-2 ("post-processing") candidate
 * verification, in plain C.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (kv-match_amd/) never links,
 * imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (DSM-fudan/KV-match, pure Java 8) ships no tests, no golden
 * vectors and no fixtures for this path, and it cannot be compiled or run in this image (no
 * JDK/JVM/Maven; every file on the path imports un-vendored hbase/slf4j artefacts).  This
 * restatement therefore follows the Java source line by line (same loop order, same running
 * sums, same early-abandon predicates, IEEE-754 binary64 without contraction) and is
 * cross-checked by independent brute-force implementations in tests/ (full-matrix banded DTW,
 * naive sliding min/max, direct z-normalised ED, self-match == 0), but no output of the Java
 * code itself was available to pin it.
 *
 * Reference paths below are relative to /root/reference/src/main/java/cn/edu/fudan/dsm/kvmatch/
 * (abbreviated K/).  Build: gcc -O2 -ffp-contract=off -fno-fast-math (Java never fuses a*b+c;
 * Math.sqrt is correctly rounded == sqrt()).
 *
 * Conventions (reference): offsets are 1-based ints; values are fp64; `shift` is
 * (lastSegment - 1) * WuList[0] with WuList[0] = 25 (K/QueryEngine.java:346).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define KVO_OK 0
#define KVO_ERR_ILLEGAL_ARGUMENT (-2) /* IllegalArgumentException, K/operator/file/TimeSeriesFileOperator.java:55-57 */
#define KVO_ERR_IO (-3)               /* IOException */
#define KVO_ERR_CAPACITY (-4)         /* caller's answer buffer too small (no reference counterpart) */
#define KVO_ERR_INDEX_OUT_OF_BOUNDS (-8) /* ArrayIndexOutOfBoundsException: DtwUtils.lowerUpperLemire with len <= r,
                                            K/utils/DtwUtils.java:81-90 (kvo_set_strict_bounds) */

#define KVO_MAX_SCAN_DATA_LENGTH 40000 /* K/NormQueryEngine.java:60, K/NormQueryEngineDtw.java:61 */
#define KVO_ROW_LENGTH 1000            /* K/common/entity/TimeSeriesNode.java:30 */

static const double KVO_INF = 1e20; /* K/utils/DtwUtils.java:24 */

/* ------------------------------------------------------------------------------------------ */
/* Answers: List<Pair<Integer, Double>> (K/QueryEngine.java:341)                                */

typedef struct {
    int32_t *off;
    double *dist;
    int64_t cap;
    int64_t cnt;
    int overflow;
} kvo_answers;

static void answers_add(kvo_answers *a, int32_t off, double dist) {
    if (a->cnt < a->cap) {
        a->off[a->cnt] = off;
        a->dist[a->cnt] = dist;
    } else {
        a->overflow = 1;
    }
    a->cnt++;
}

/* Double.compare semantics (used by Comparator.comparing(Pair::getSecond), K/QueryEngine.java:373). */
static int double_compare(double a, double b) {
    if (a < b) return -1;
    if (a > b) return 1;
    int64_t x, y;
    memcpy(&x, &a, 8);
    memcpy(&y, &b, 8);
    /* equal, or at least one NaN: Java orders by raw bits with NaN canonicalised above +inf */
    if (a != a) x = 0x7ff8000000000000LL;
    if (b != b) y = 0x7ff8000000000000LL;
    return (x == y) ? 0 : (x < y ? -1 : 1);
}

/* Stable merge sort of (off, dist) by dist -- List.sort is a stable TimSort; with a consistent
 * comparator every stable sort yields the same permutation (K/QueryEngine.java:373). */
static void stable_sort_answers(int32_t *off, double *dist, int64_t n) {
    if (n < 2) return;
    int32_t *toff = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    double *tdist = (double *)malloc((size_t)n * sizeof(double));
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n;
            int64_t hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                if (double_compare(dist[j], dist[i]) < 0) {
                    toff[o] = off[j];
                    tdist[o++] = dist[j++];
                } else {
                    toff[o] = off[i];
                    tdist[o++] = dist[i++];
                }
            }
            while (i < mid) {
                toff[o] = off[i];
                tdist[o++] = dist[i++];
            }
            while (j < hi) {
                toff[o] = off[j];
                tdist[o++] = dist[j++];
            }
        }
        memcpy(off, toff, (size_t)n * sizeof(int32_t));
        memcpy(dist, tdist, (size_t)n * sizeof(double));
    }
    free(toff);
    free(tdist);
}

void kvo_sort_answers(int32_t *off, double *dist, int64_t n) { stable_sort_answers(off, dist, n); }

/* ------------------------------------------------------------------------------------------ */
/* Data access: TimeSeriesOperator.readTimeSeries(left, length)                                 */
/* (K/operator/TimeSeriesOperator.java:38; file backend K/operator/file/TimeSeriesFileOperator.java:54-96). */
/* The series is held in memory, decoded once; `series[p-1]` is the value at 1-based offset p.   */

static int read_time_series(const double *series, int n, int left, int length, const double **out) {
    if (left < 1 || (int64_t)left + length - 1 > n || length < 1) return KVO_ERR_ILLEGAL_ARGUMENT;
    *out = series + (left - 1);
    return KVO_OK;
}

/* Big-endian IEEE-754 decode: Bytes.toDouble (hbase-common 1.1.5), call site
 * K/common/entity/TimeSeriesNode.java:46-52. */
void kvo_decode_be64(const uint8_t *bytes, int64_t count, double *out) {
    for (int64_t i = 0; i < count; i++) {
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v = (v << 8) | bytes[i * 8 + b];
        memcpy(&out[i], &v, 8);
    }
}

/* Encode (Bytes.toBytes(double), K/common/entity/TimeSeriesNode.java:38-44; DataOutputStream.writeDouble
 * K/DataGenerator.java:111). */
void kvo_encode_be64(const double *in, int64_t count, uint8_t *bytes) {
    for (int64_t i = 0; i < count; i++) {
        uint64_t v;
        memcpy(&v, &in[i], 8);
        for (int b = 7; b >= 0; b--) {
            bytes[i * 8 + b] = (uint8_t)(v & 0xff);
            v >>= 8;
        }
    }
}

/* readTimeSeries straight from files/data-{N}, following the node-granular addressing of
 * K/operator/file/TimeSeriesFileOperator.java:54-96 (whole ROW_LENGTH-value nodes are read and
 * decoded, then trimmed).  A short final node is tolerated (n not a multiple of 1000). */
int kvo_read_time_series_file(const char *path, int n, int left, int length, double *out) {
    if (left < 1 || (int64_t)left + length - 1 > n || length < 1) return KVO_ERR_ILLEGAL_ARGUMENT;
    int nodeBytes = KVO_ROW_LENGTH * 8;
    int begin = (left - 1) / KVO_ROW_LENGTH * KVO_ROW_LENGTH + 1;
    int end = (left + length - 2) / KVO_ROW_LENGTH * KVO_ROW_LENGTH + 1;
    int64_t pos = (int64_t)((left - 1) / KVO_ROW_LENGTH) * (int64_t)nodeBytes;
    int64_t lengthOfBytes = (int64_t)(end - begin + KVO_ROW_LENGTH) * 8;
    FILE *f = fopen(path, "rb");
    if (!f) return KVO_ERR_IO;
    uint8_t *bytes = (uint8_t *)calloc((size_t)lengthOfBytes, 1);
    if (fseek(f, (long)pos, SEEK_SET) != 0) {
        fclose(f);
        free(bytes);
        return KVO_ERR_IO;
    }
    size_t got = fread(bytes, 1, (size_t)lengthOfBytes, f);
    fclose(f);
    (void)got;
    int64_t cnt = 0;
    for (int64_t i = 0; i < lengthOfBytes; i += nodeBytes) {
        double node[KVO_ROW_LENGTH];
        kvo_decode_be64(bytes + i, KVO_ROW_LENGTH, node);
        if (i == 0) {
            int offset = (left - 1) % KVO_ROW_LENGTH;
            int hi = offset + length < KVO_ROW_LENGTH ? offset + length : KVO_ROW_LENGTH;
            for (int j = offset; j < hi; j++) out[cnt++] = node[j];
        } else if (i == lengthOfBytes - nodeBytes) {
            int offset = (left + length - 2) % KVO_ROW_LENGTH;
            for (int j = 0; j <= offset; j++) out[cnt++] = node[j];
        } else {
            for (int j = 0; j < KVO_ROW_LENGTH; j++) out[cnt++] = node[j];
        }
    }
    free(bytes);
    return cnt == length ? KVO_OK : KVO_ERR_IO;
}

/* ------------------------------------------------------------------------------------------ */
/* CircularArray (K/common/CircularArray.java:21-110): int deque used by the envelope.           */

typedef struct {
    int *dq;
    int size, capacity, f, r;
} circ;

static void circ_init(circ *c, int capacity) {
    c->capacity = capacity;
    c->size = 0;
    c->dq = (int *)malloc((size_t)capacity * sizeof(int));
    c->f = 0;
    c->r = capacity - 1;
}
static void circ_free(circ *c) { free(c->dq); }
static void circ_push_back(circ *c, int v) {
    c->dq[c->r] = v;
    c->r--;
    if (c->r < 0) c->r = c->capacity - 1;
    c->size++;
}
static void circ_pop_front(circ *c) {
    c->f--;
    if (c->f < 0) c->f = c->capacity - 1;
    c->size--;
}
static void circ_pop_back(circ *c) {
    c->r = (c->r + 1) % c->capacity;
    c->size--;
}
static int circ_front(const circ *c) {
    int aux = c->f - 1;
    if (aux < 0) aux = c->capacity - 1;
    return c->dq[aux];
}
static int circ_back(const circ *c) {
    int aux = (c->r + 1) % c->capacity;
    return c->dq[aux];
}
static int circ_is_empty(const circ *c) { return c->size == 0; }

/* ------------------------------------------------------------------------------------------ */
/* DtwUtils (K/utils/DtwUtils.java)                                                              */

static double dmin(double a, double b) { return (a < b) ? a : b; }     /* :34-36 */
static double ddist(double v1, double v2) { return (v1 - v2) * (v1 - v2); } /* :38-40 */
static int imax(int a, int b) { return (a > b) ? a : b; }               /* :26-28 */
static int imin(int a, int b) { return (a < b) ? a : b; }               /* :30-32 */

/* lowerUpperLemire, array and List<Double> overloads are the same algorithm
 * (K/utils/DtwUtils.java:50-91 and :93-134). */
/* `u[i - r - 1]` of the closing loop (:81-90) starts at i = len: with len <= r the index is negative and Java throws
 * ArrayIndexOutOfBoundsException -- the engines never catch it, the query dies.  Strict (default): the callers below return
 * KVO_ERR_INDEX_OUT_OF_BOUNDS at that point.  Not strict: the stores outside the arrays are skipped ("natural extension":
 * a window clipped at the ends of the array) and the query goes on. */
static int kvo_strict_bounds = 1;
static int kvo_oob = 0;
void kvo_set_strict_bounds(int on) { kvo_strict_bounds = on ? 1 : 0; }
static int kvo_take_oob(void) { int v = kvo_oob; kvo_oob = 0; return v; }

void kvo_lower_upper_lemire(const double *t, int len, int r, double *l, double *u) {
    if (kvo_strict_bounds && len <= r) {
        kvo_oob = 1;
        return;
    }
    circ du, dl;
    circ_init(&du, 2 * r + 2);
    circ_init(&dl, 2 * r + 2);

    circ_push_back(&du, 0);
    circ_push_back(&dl, 0);

    for (int i = 1; i < len; i++) {
        if (i > r) {
            u[i - r - 1] = t[circ_front(&du)];
            l[i - r - 1] = t[circ_front(&dl)];
        }
        if (t[i] > t[i - 1]) {
            circ_pop_back(&du);
            while (!circ_is_empty(&du) && t[i] > t[circ_back(&du)]) {
                circ_pop_back(&du);
            }
        } else {
            circ_pop_back(&dl);
            while (!circ_is_empty(&dl) && t[i] < t[circ_back(&dl)]) {
                circ_pop_back(&dl);
            }
        }
        circ_push_back(&du, i);
        circ_push_back(&dl, i);
        if (i == 2 * r + 1 + circ_front(&du)) {
            circ_pop_front(&du);
        } else if (i == 2 * r + 1 + circ_front(&dl)) {
            circ_pop_front(&dl);
        }
    }
    for (int i = len; i < len + r + 1; i++) {
        /* (not strict only: Java throws ArrayIndexOutOfBounds for i - r - 1 < 0, i.e. len <= r) */
        if (i - r - 1 >= 0) {
            u[i - r - 1] = t[circ_front(&du)];
            l[i - r - 1] = t[circ_front(&dl)];
        }
        if (i - circ_front(&du) >= 2 * r + 1) {
            circ_pop_front(&du);
        }
        if (i - circ_front(&dl) >= 2 * r + 1) {
            circ_pop_front(&dl);
        }
    }
    circ_free(&du);
    circ_free(&dl);
}

/* lbKimHierarchy (K/utils/DtwUtils.java:149-189) */
double kvo_lb_kim_hierarchy(const double *t, const double *q, int j, int len, double mean, double std,
                            double bsf) {
    double d, lb;
    double x0 = (t[j] - mean) / std;
    double y0 = (t[(len - 1 + j)] - mean) / std;
    lb = ddist(x0, q[0]) + ddist(y0, q[len - 1]);
    if (lb >= bsf) return lb;

    double x1 = (t[(j + 1)] - mean) / std;
    d = dmin(ddist(x1, q[0]), ddist(x0, q[1]));
    d = dmin(d, ddist(x1, q[1]));
    lb += d;
    if (lb >= bsf) return lb;

    double y1 = (t[(len - 2 + j)] - mean) / std;
    d = dmin(ddist(y1, q[len - 1]), ddist(y0, q[len - 2]));
    d = dmin(d, ddist(y1, q[len - 2]));
    lb += d;
    if (lb >= bsf) return lb;

    double x2 = (t[(j + 2)] - mean) / std;
    d = dmin(ddist(x0, q[2]), ddist(x1, q[2]));
    d = dmin(d, ddist(x2, q[2]));
    d = dmin(d, ddist(x2, q[1]));
    d = dmin(d, ddist(x2, q[0]));
    lb += d;
    if (lb >= bsf) return lb;

    double y2 = (t[(len - 3 + j)] - mean) / std;
    d = dmin(ddist(y0, q[len - 3]), ddist(y1, q[len - 3]));
    d = dmin(d, ddist(y2, q[len - 3]));
    d = dmin(d, ddist(y2, q[len - 2]));
    d = dmin(d, ddist(y2, q[len - 1]));
    lb += d;

    return lb;
}

/* lbKeoghCumulative (K/utils/DtwUtils.java:206-222) */
double kvo_lb_keogh_cumulative(const int *order, const double *t, const double *uo, const double *lo,
                               double *cb, int j, int len, double mean, double std, double best_so_far) {
    double lb = 0;
    double x, d;

    for (int i = 0; i < len && lb < best_so_far; i++) {
        x = (t[(order[i] + j)] - mean) / std;
        d = 0;
        if (x > uo[i]) {
            d = ddist(x, uo[i]);
        } else if (x < lo[i]) {
            d = ddist(x, lo[i]);
        }
        lb += d;
        cb[order[i]] = d;
    }
    return lb;
}

/* lbKeoghDataCumulative (K/utils/DtwUtils.java:238-257) */
double kvo_lb_keogh_data_cumulative(const int *order, const double *qo, double *cb, int I, const double *l,
                                    const double *u, int len, double mean, double std, double best_so_far) {
    double lb = 0;
    double uu, ll, d;

    for (int i = 0; i < len && lb < best_so_far; i++) {
        uu = (u[order[i] + I] - mean) / std;
        ll = (l[order[i] + I] - mean) / std;
        d = 0;
        if (qo[i] > uu) {
            d = ddist(qo[i], uu);
        } else {
            if (qo[i] < ll) {
                d = ddist(qo[i], ll);
            }
        }
        lb += d;
        cb[order[i]] = d;
    }
    return lb;
}

/* dtw (K/utils/DtwUtils.java:269-337).  `cells`, when non-NULL, accumulates the number of DP
 * cells evaluated (instrumentation only; used by the bench's work accounting). */
double kvo_dtw(const double *A, const double *B, const double *cb, int m, int r, double bsf, int64_t *cells) {
    double *cost, *cost_prev, *cost_tmp;
    int i, j, k;
    double x, y, z, min_cost;

    cost = (double *)malloc((size_t)(2 * r + 1) * sizeof(double));
    for (k = 0; k < 2 * r + 1; k++) cost[k] = KVO_INF;
    cost_prev = (double *)malloc((size_t)(2 * r + 1) * sizeof(double));
    for (k = 0; k < 2 * r + 1; k++) cost_prev[k] = KVO_INF;

    k = 0;
    for (i = 0; i < m; i++) {
        k = imax(0, r - i);
        min_cost = KVO_INF;

        for (j = imax(0, i - r); j <= imin(m - 1, i + r); j++, k++) {
            if (cells) (*cells)++;
            if ((i == 0) && (j == 0)) {
                cost[k] = ddist(A[0], B[0]);
                min_cost = cost[k];
                continue;
            }

            if ((j - 1 < 0) || (k - 1 < 0)) {
                y = KVO_INF;
            } else {
                y = cost[k - 1];
            }
            if ((i - 1 < 0) || (k + 1 > 2 * r)) {
                x = KVO_INF;
            } else {
                x = cost_prev[k + 1];
            }
            if ((i - 1 < 0) || (j - 1 < 0)) {
                z = KVO_INF;
            } else {
                z = cost_prev[k];
            }

            cost[k] = dmin(dmin(x, y), z) + ddist(A[i], B[j]);

            if (cost[k] < min_cost) {
                min_cost = cost[k];
            }
        }

        if (i + r < m - 1 && min_cost + cb[i + r + 1] >= bsf) {
            double ret = min_cost + cb[i + r + 1];
            free(cost);
            free(cost_prev);
            return ret;
        }

        cost_tmp = cost;
        cost = cost_prev;
        cost_prev = cost_tmp;
    }
    k--;

    double final_dtw = cost_prev[k];
    free(cost);
    free(cost_prev);
    return final_dtw;
}

/* ------------------------------------------------------------------------------------------ */
/* Statistics out: [0]=cntCandidate (statistics[3], K/QueryEngine.java:369), [1]=#answers,       */
/* then instrumentation (no reference counterpart): [2]=#windows evaluated, [3]=#passed (alpha,   */
/* beta) filter, [4]=#passed lbKim, [5]=#passed lbKeogh, [6]=#passed lbKeogh2 (= #dtw calls),     */
/* [7]=#DP cells evaluated, [8]=#ED terms evaluated.                                              */
#define KVO_NSTAT 9

/* ------------------------------------------------------------------------------------------ */
/* RSM-ED phase 2 (K/QueryEngine.java:338-363, sort :373).                                       */
/* no_abandon != 0 disables the `distance <= epsilon*epsilon` loop guard (measurement only: the  */
/* answers are identical because answers always complete the loop).                              */
int64_t kvo_query_rsm_ed(const double *series, int n, const double *queryData, int length, const int32_t *left,
                         const int32_t *right, int k, int shift, double epsilon, int no_abandon, int32_t *out_off,
                         double *out_dist, int64_t cap, int64_t *stats) {
    kvo_answers answers = {out_off, out_dist, cap, 0, 0};
    int64_t cntCandidate = 0, windows = 0, terms = 0;
    for (int p = 0; p < k; p++) {
        cntCandidate += right[p] - left[p] + 1;

        int begin = left[p] - shift;
        int end = right[p] - shift + length - 1;
        if (begin < 1) begin = 1;
        if (end > n) end = n;
        const double *data;
        int dataSize = end - begin + 1;
        int rc = read_time_series(series, n, begin, dataSize, &data);
        if (rc != KVO_OK) return rc;

        for (int i = 0; i + length - 1 < dataSize; i++) {
            double distance = 0;
            windows++;
            int j;
            for (j = 0; j < length && (no_abandon || distance <= epsilon * epsilon); j++) {
                distance += (data[i + j] - queryData[j]) * (data[i + j] - queryData[j]);
            }
            terms += j;
            if (distance <= epsilon * epsilon) {
                answers_add(&answers, begin + i, sqrt(distance));
            }
        }
    }
    if (answers.overflow) return KVO_ERR_CAPACITY;
    stable_sort_answers(out_off, out_dist, answers.cnt);
    if (stats) {
        memset(stats, 0, KVO_NSTAT * sizeof(int64_t));
        stats[0] = cntCandidate;
        stats[1] = answers.cnt;
        stats[2] = windows;
        stats[8] = terms;
    }
    return answers.cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* Index (K/common/Index.java:18-28) + the two query reorderings.                                */

typedef struct {
    double value;
    int index;
} kvo_index;

/* Stable merge sort with a caller-supplied three-way comparator. */
typedef int (*index_cmp)(const kvo_index *, const kvo_index *);
static void stable_sort_index(kvo_index *a, int n, index_cmp cmp) {
    if (n < 2) return;
    kvo_index *tmp = (kvo_index *)malloc((size_t)n * sizeof(kvo_index));
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n;
            int hi = lo + 2 * w < n ? lo + 2 * w : n;
            int i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                if (cmp(&a[j], &a[i]) < 0)
                    tmp[o++] = a[j++];
                else
                    tmp[o++] = a[i++];
            }
            while (i < mid) tmp[o++] = a[i++];
            while (j < hi) tmp[o++] = a[j++];
        }
        memcpy(a, tmp, (size_t)n * sizeof(kvo_index));
    }
    free(tmp);
}

/* (o1, o2) -> Double.compare(Math.abs(o2.value), Math.abs(o1.value))  (K/NormQueryEngine.java:449) */
static int cmp_abs_desc(const kvo_index *o1, const kvo_index *o2) {
    return double_compare(fabs(o2->value), fabs(o1->value));
}

/* (o1, o2) -> (int) (Math.abs(o2.value) - Math.abs(o1.value))  (K/NormQueryEngineDtw.java:475-478).
 * The truncation makes this comparator inconsistent ("equal" whenever the |values| are less than 1
 * apart, not transitive), so the permutation Arrays.sort returns is a property of the JDK's sorting
 * algorithm, and the sort may end in IllegalArgumentException.  kvo_java_timsort below restates it. */
static int cmp_abs_desc_trunc(const kvo_index *o1, const kvo_index *o2) {
    return (int)(fabs(o2->value) - fabs(o1->value));
}

/* ------------------------------------------------------------------------------------------ */
/* java.util.TimSort as Arrays.sort(T[], Comparator) runs it: TimSort.sort(a, 0, a.length, c, null, 0, 0).
 * THIRD-PARTY DEPENDENCY absent from /root/reference: the JDK (pom.xml targets Java 1.8).  Restated from
 * the published OpenJDK 8 source of java/util/TimSort.java (sort, countRunAndMakeAscending, binarySort,
 * minRunLength, mergeCollapse [the JDK 8 form], mergeForceCollapse, mergeAt, gallopLeft, gallopRight,
 * mergeLo, mergeHi).  No JDK is installed here, so this restatement is not checked against a JVM
 * ("parity unpinned", like the rest of this file); tests/java_transliteration.py holds a second
 * restatement and tests/test_transliteration.py compares the two.  What does not depend on the restatement
 * being right in every detail: an array in which compare(a[i], a[i-1]) >= 0 for all i is one run and is
 * returned unchanged -- under the comparator above that is every query whose |zQ| never rises by >= 1
 * between neighbours, i.e. most smooth queries.
 * Returns 0, or -1 where Java throws IllegalArgumentException("Comparison method violates its general
 * contract!") (the array is then left partially merged, as in Java). */
#define TS_MIN_MERGE 32
#define TS_MIN_GALLOP 7

typedef struct {
    kvo_index *a;
    index_cmp c;
    int minGallop;
    kvo_index *tmp;
    int stackSize;
    int runBase[64], runLen[64];
} kvo_timsort;

static int ts_count_run_and_make_ascending(kvo_index *a, int lo, int hi, index_cmp c) {
    int runHi = lo + 1;
    if (runHi == hi) return 1;
    if (c(&a[runHi++], &a[lo]) < 0) { /* descending */
        while (runHi < hi && c(&a[runHi], &a[runHi - 1]) < 0) runHi++;
        for (int i = lo, j = runHi - 1; i < j; i++, j--) { /* reverseRange */
            kvo_index t = a[i];
            a[i] = a[j];
            a[j] = t;
        }
    } else { /* ascending */
        while (runHi < hi && c(&a[runHi], &a[runHi - 1]) >= 0) runHi++;
    }
    return runHi - lo;
}

static void ts_binary_sort(kvo_index *a, int lo, int hi, int start, index_cmp c) {
    if (start == lo) start++;
    for (; start < hi; start++) {
        kvo_index pivot = a[start];
        int left = lo, right = start;
        while (left < right) {
            int mid = (int)(((unsigned)left + (unsigned)right) >> 1);
            if (c(&pivot, &a[mid]) < 0)
                right = mid;
            else
                left = mid + 1;
        }
        memmove(&a[left + 1], &a[left], (size_t)(start - left) * sizeof(kvo_index));
        a[left] = pivot;
    }
}

static int ts_min_run_length(int n) {
    int r = 0;
    while (n >= TS_MIN_MERGE) {
        r |= (n & 1);
        n >>= 1;
    }
    return n + r;
}

static int ts_gallop_left(const kvo_index *key, const kvo_index *a, int base, int len, int hint, index_cmp c) {
    int lastOfs = 0, ofs = 1;
    if (c(key, &a[base + hint]) > 0) {
        int maxOfs = len - hint;
        while (ofs < maxOfs && c(key, &a[base + hint + ofs]) > 0) {
            lastOfs = ofs;
            ofs = (ofs << 1) + 1;
            if (ofs <= 0) ofs = maxOfs;
        }
        if (ofs > maxOfs) ofs = maxOfs;
        lastOfs += hint;
        ofs += hint;
    } else {
        int maxOfs = hint + 1;
        while (ofs < maxOfs && c(key, &a[base + hint - ofs]) <= 0) {
            lastOfs = ofs;
            ofs = (ofs << 1) + 1;
            if (ofs <= 0) ofs = maxOfs;
        }
        if (ofs > maxOfs) ofs = maxOfs;
        int t = lastOfs;
        lastOfs = hint - ofs;
        ofs = hint - t;
    }
    lastOfs++;
    while (lastOfs < ofs) {
        int m = lastOfs + ((ofs - lastOfs) >> 1);
        if (c(key, &a[base + m]) > 0)
            lastOfs = m + 1;
        else
            ofs = m;
    }
    return ofs;
}

static int ts_gallop_right(const kvo_index *key, const kvo_index *a, int base, int len, int hint, index_cmp c) {
    int ofs = 1, lastOfs = 0;
    if (c(key, &a[base + hint]) < 0) {
        int maxOfs = hint + 1;
        while (ofs < maxOfs && c(key, &a[base + hint - ofs]) < 0) {
            lastOfs = ofs;
            ofs = (ofs << 1) + 1;
            if (ofs <= 0) ofs = maxOfs;
        }
        if (ofs > maxOfs) ofs = maxOfs;
        int t = lastOfs;
        lastOfs = hint - ofs;
        ofs = hint - t;
    } else {
        int maxOfs = len - hint;
        while (ofs < maxOfs && c(key, &a[base + hint + ofs]) >= 0) {
            lastOfs = ofs;
            ofs = (ofs << 1) + 1;
            if (ofs <= 0) ofs = maxOfs;
        }
        if (ofs > maxOfs) ofs = maxOfs;
        lastOfs += hint;
        ofs += hint;
    }
    lastOfs++;
    while (lastOfs < ofs) {
        int m = lastOfs + ((ofs - lastOfs) >> 1);
        if (c(key, &a[base + m]) < 0)
            ofs = m;
        else
            lastOfs = m + 1;
    }
    return ofs;
}

static int ts_merge_lo(kvo_timsort *ts, int base1, int len1, int base2, int len2) {
    kvo_index *a = ts->a, *tmp = ts->tmp;
    index_cmp c = ts->c;
    int cursor1 = 0, cursor2 = base2, dest = base1;
    memcpy(tmp, &a[base1], (size_t)len1 * sizeof(kvo_index));
    a[dest++] = a[cursor2++];
    if (--len2 == 0) {
        memcpy(&a[dest], &tmp[cursor1], (size_t)len1 * sizeof(kvo_index));
        return 0;
    }
    if (len1 == 1) {
        memmove(&a[dest], &a[cursor2], (size_t)len2 * sizeof(kvo_index));
        a[dest + len2] = tmp[cursor1];
        return 0;
    }
    int minGallop = ts->minGallop;
    for (;;) {
        int count1 = 0, count2 = 0;
        do {
            if (c(&a[cursor2], &tmp[cursor1]) < 0) {
                a[dest++] = a[cursor2++];
                count2++;
                count1 = 0;
                if (--len2 == 0) goto outer;
            } else {
                a[dest++] = tmp[cursor1++];
                count1++;
                count2 = 0;
                if (--len1 == 1) goto outer;
            }
        } while ((count1 | count2) < minGallop);
        do {
            count1 = ts_gallop_right(&a[cursor2], tmp, cursor1, len1, 0, c);
            if (count1 != 0) {
                memcpy(&a[dest], &tmp[cursor1], (size_t)count1 * sizeof(kvo_index));
                dest += count1;
                cursor1 += count1;
                len1 -= count1;
                if (len1 <= 1) goto outer;
            }
            a[dest++] = a[cursor2++];
            if (--len2 == 0) goto outer;
            count2 = ts_gallop_left(&tmp[cursor1], a, cursor2, len2, 0, c);
            if (count2 != 0) {
                memmove(&a[dest], &a[cursor2], (size_t)count2 * sizeof(kvo_index));
                dest += count2;
                cursor2 += count2;
                len2 -= count2;
                if (len2 == 0) goto outer;
            }
            a[dest++] = tmp[cursor1++];
            if (--len1 == 1) goto outer;
            minGallop--;
        } while ((count1 >= TS_MIN_GALLOP) | (count2 >= TS_MIN_GALLOP));
        if (minGallop < 0) minGallop = 0;
        minGallop += 2;
    }
outer:
    ts->minGallop = minGallop < 1 ? 1 : minGallop;
    if (len1 == 1) {
        memmove(&a[dest], &a[cursor2], (size_t)len2 * sizeof(kvo_index));
        a[dest + len2] = tmp[cursor1];
    } else if (len1 == 0) {
        return -1; /* "Comparison method violates its general contract!" */
    } else {
        memcpy(&a[dest], &tmp[cursor1], (size_t)len1 * sizeof(kvo_index));
    }
    return 0;
}

static int ts_merge_hi(kvo_timsort *ts, int base1, int len1, int base2, int len2) {
    kvo_index *a = ts->a, *tmp = ts->tmp;
    index_cmp c = ts->c;
    const int tmpBase = 0;
    memcpy(tmp, &a[base2], (size_t)len2 * sizeof(kvo_index));
    int cursor1 = base1 + len1 - 1, cursor2 = tmpBase + len2 - 1, dest = base2 + len2 - 1;
    a[dest--] = a[cursor1--];
    if (--len1 == 0) {
        memcpy(&a[dest - (len2 - 1)], &tmp[tmpBase], (size_t)len2 * sizeof(kvo_index));
        return 0;
    }
    if (len2 == 1) {
        dest -= len1;
        cursor1 -= len1;
        memmove(&a[dest + 1], &a[cursor1 + 1], (size_t)len1 * sizeof(kvo_index));
        a[dest] = tmp[cursor2];
        return 0;
    }
    int minGallop = ts->minGallop;
    for (;;) {
        int count1 = 0, count2 = 0;
        do {
            if (c(&tmp[cursor2], &a[cursor1]) < 0) {
                a[dest--] = a[cursor1--];
                count1++;
                count2 = 0;
                if (--len1 == 0) goto outer;
            } else {
                a[dest--] = tmp[cursor2--];
                count2++;
                count1 = 0;
                if (--len2 == 1) goto outer;
            }
        } while ((count1 | count2) < minGallop);
        do {
            count1 = len1 - ts_gallop_right(&tmp[cursor2], a, base1, len1, len1 - 1, c);
            if (count1 != 0) {
                dest -= count1;
                cursor1 -= count1;
                len1 -= count1;
                memmove(&a[dest + 1], &a[cursor1 + 1], (size_t)count1 * sizeof(kvo_index));
                if (len1 == 0) goto outer;
            }
            a[dest--] = tmp[cursor2--];
            if (--len2 == 1) goto outer;
            count2 = len2 - ts_gallop_left(&a[cursor1], tmp, tmpBase, len2, len2 - 1, c);
            if (count2 != 0) {
                dest -= count2;
                cursor2 -= count2;
                len2 -= count2;
                memcpy(&a[dest + 1], &tmp[cursor2 + 1], (size_t)count2 * sizeof(kvo_index));
                if (len2 <= 1) goto outer;
            }
            a[dest--] = a[cursor1--];
            if (--len1 == 0) goto outer;
            minGallop--;
        } while ((count1 >= TS_MIN_GALLOP) | (count2 >= TS_MIN_GALLOP));
        if (minGallop < 0) minGallop = 0;
        minGallop += 2;
    }
outer:
    ts->minGallop = minGallop < 1 ? 1 : minGallop;
    if (len2 == 1) {
        dest -= len1;
        cursor1 -= len1;
        memmove(&a[dest + 1], &a[cursor1 + 1], (size_t)len1 * sizeof(kvo_index));
        a[dest] = tmp[cursor2];
    } else if (len2 == 0) {
        return -1; /* "Comparison method violates its general contract!" */
    } else {
        memcpy(&a[dest - (len2 - 1)], &tmp[tmpBase], (size_t)len2 * sizeof(kvo_index));
    }
    return 0;
}

static int ts_merge_at(kvo_timsort *ts, int i) {
    kvo_index *a = ts->a;
    int base1 = ts->runBase[i], len1 = ts->runLen[i];
    int base2 = ts->runBase[i + 1], len2 = ts->runLen[i + 1];
    ts->runLen[i] = len1 + len2;
    if (i == ts->stackSize - 3) {
        ts->runBase[i + 1] = ts->runBase[i + 2];
        ts->runLen[i + 1] = ts->runLen[i + 2];
    }
    ts->stackSize--;
    int k = ts_gallop_right(&a[base2], a, base1, len1, 0, ts->c);
    base1 += k;
    len1 -= k;
    if (len1 == 0) return 0;
    len2 = ts_gallop_left(&a[base1 + len1 - 1], a, base2, len2, len2 - 1, ts->c);
    if (len2 == 0) return 0;
    return len1 <= len2 ? ts_merge_lo(ts, base1, len1, base2, len2) : ts_merge_hi(ts, base1, len1, base2, len2);
}

static int ts_merge_collapse(kvo_timsort *ts) {
    while (ts->stackSize > 1) {
        int n = ts->stackSize - 2;
        if (n > 0 && ts->runLen[n - 1] <= ts->runLen[n] + ts->runLen[n + 1]) {
            if (ts->runLen[n - 1] < ts->runLen[n + 1]) n--;
            if (ts_merge_at(ts, n)) return -1;
        } else if (ts->runLen[n] <= ts->runLen[n + 1]) {
            if (ts_merge_at(ts, n)) return -1;
        } else {
            break;
        }
    }
    return 0;
}

static int ts_merge_force_collapse(kvo_timsort *ts) {
    while (ts->stackSize > 1) {
        int n = ts->stackSize - 2;
        if (n > 0 && ts->runLen[n - 1] < ts->runLen[n + 1]) n--;
        if (ts_merge_at(ts, n)) return -1;
    }
    return 0;
}

static int kvo_java_timsort(kvo_index *a, int n, index_cmp c) {
    int lo = 0, hi = n, nRemaining = n;
    if (nRemaining < 2) return 0;
    if (nRemaining < TS_MIN_MERGE) {
        int initRunLen = ts_count_run_and_make_ascending(a, lo, hi, c);
        ts_binary_sort(a, lo, hi, lo + initRunLen, c);
        return 0;
    }
    kvo_timsort ts;
    ts.a = a;
    ts.c = c;
    ts.minGallop = TS_MIN_GALLOP;
    ts.stackSize = 0;
    ts.tmp = (kvo_index *)malloc((size_t)n * sizeof(kvo_index)); /* Java grows tmp on demand; the size is not observable */
    int minRun = ts_min_run_length(nRemaining), rc = 0;
    do {
        int runLen = ts_count_run_and_make_ascending(a, lo, hi, c);
        if (runLen < minRun) {
            int force = nRemaining <= minRun ? nRemaining : minRun;
            ts_binary_sort(a, lo, lo + force, lo + runLen, c);
            runLen = force;
        }
        if (ts.stackSize == 64) { /* unreachable below 2^31 elements; Java's stack (49 entries) would overflow first */
            rc = -1;
            break;
        }
        ts.runBase[ts.stackSize] = lo;
        ts.runLen[ts.stackSize] = runLen;
        ts.stackSize++;
        if ((rc = ts_merge_collapse(&ts)) != 0) break;
        lo += runLen;
        nRemaining -= runLen;
    } while (nRemaining != 0);
    if (rc == 0) rc = ts_merge_force_collapse(&ts);
    free(ts.tmp);
    return rc;
}

/* TEST SWITCH, not reference behaviour: 1 = where the sort above throws, carry on with the array in index order (the
 * library's KVM_OPT_SORT_VIOLATION = 1), so that the verification arithmetic can still be compared on such queries. */
static int kvo_lenient_sort = 0;
void kvo_set_lenient_sort(int on) { kvo_lenient_sort = on ? 1 : 0; }

/* The permutation NormQueryEngineDtw sorts its query into (K/NormQueryEngineDtw.java:470-484): order[i] = the
 * index of the i-th element of tmpQ after Arrays.sort.  Returns KVO_OK or KVO_ERR_ILLEGAL_ARGUMENT where the
 * sort throws (query() then ends in that IllegalArgumentException: nothing catches it). */
int kvo_cnsm_dtw_order(const double *zQ, int queryLength, int *order) {
    kvo_index *tmpQ = (kvo_index *)malloc((size_t)(queryLength > 0 ? queryLength : 1) * sizeof(kvo_index));
    for (int i = 0; i < queryLength; i++) {
        tmpQ[i].value = zQ[i];
        tmpQ[i].index = i;
    }
    int rc = kvo_java_timsort(tmpQ, queryLength, cmp_abs_desc_trunc);
    for (int i = 0; i < queryLength; i++) order[i] = tmpQ[i].index;
    free(tmpQ);
    return rc ? KVO_ERR_ILLEGAL_ARGUMENT : KVO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* cNSM-ED phase 2 (K/NormQueryEngine.java:192-198 statistics, :432-528 phase 2).               */
int64_t kvo_query_cnsm_ed(const double *series, int n, const double *queryData, int queryLength,
                          const int32_t *left, const int32_t *right, int k, int shift, double epsilon, double alpha,
                          double beta, int no_abandon, int32_t *out_off, double *out_dist, int64_t cap,
                          int64_t *stats) {
    /* GUARD (not in the Java code): the chunk loop below indexes the fetched chunk with `left - begin`.  The engines only
     * ever pass the list sortAndMergeIntervals leaves (sorted by left, K/QueryEngine.java:664-693); on a list that is not
     * sorted by left Java ends in the range check of readTimeSeries or in an IndexOutOfBoundsException, depending on the
     * list, and C would read outside the chunk.  Reported as the IllegalArgument case (the library does the same). */
    for (int p = 1; p < k; p++)
        if (left[p] < left[p - 1]) return KVO_ERR_ILLEGAL_ARGUMENT;
    kvo_answers answers = {out_off, out_dist, cap, 0, 0};
    int64_t cntCandidate = 0, windows = 0, passed = 0, terms = 0;

    /* Phase 0 (K/NormQueryEngine.java:192-198) */
    double ex = 0, ex2 = 0;
    for (int i = 0; i < queryLength; i++) {
        double value = queryData[i];
        ex += value;
        ex2 += value * value;
    }
    double meanQ = ex / queryLength;
    double stdQ = sqrt(ex2 / queryLength - meanQ * meanQ);

    /* z-normalise + reorder (:438-452) */
    double *zQ = (double *)calloc((size_t)queryLength, sizeof(double));
    for (int i = 0; i < queryLength; i++) zQ[i] = (queryData[i] - meanQ) / stdQ;
    int *order = (int *)malloc((size_t)queryLength * sizeof(int));
    kvo_index *Q_tmp = (kvo_index *)malloc((size_t)queryLength * sizeof(kvo_index));
    for (int i = 0; i < queryLength; i++) {
        Q_tmp[i].value = zQ[i];
        Q_tmp[i].index = i;
    }
    stable_sort_index(Q_tmp, queryLength, cmp_abs_desc);
    for (int i = 0; i < queryLength; i++) {
        zQ[i] = Q_tmp[i].value;
        order[i] = Q_tmp[i].index;
    }
    double *T = (double *)malloc((size_t)2 * queryLength * sizeof(double));
    int rc = KVO_OK;

    int idx = 0;
    while (idx < k) {
        int beginIdx = idx, endIdx = idx;

        int begin = left[idx] - shift;
        int end = right[idx] - shift + queryLength - 1;
        if (begin < 1) begin = 1;
        int length = end - begin + 1;
        idx++;

        while (idx < k) {
            begin = left[idx] - shift;
            int newLength = length + begin - end - 1;
            end = right[idx] - shift + queryLength - 1;
            if (end > n) end = n;
            newLength += end - begin + 1;
            if (newLength > KVO_MAX_SCAN_DATA_LENGTH) break;
            endIdx = idx;
            length = newLength;
            idx++;
        }

        begin = left[beginIdx] - shift;
        end = right[endIdx] - shift + queryLength - 1;
        if (begin < 1) begin = 1;
        if (end > n) end = n;
        const double *data;
        int dataSize = end - begin + 1;
        rc = read_time_series(series, n, begin, dataSize, &data);
        if (rc != KVO_OK) break;

        for (int idx1 = beginIdx; idx1 <= endIdx; idx1++) {
            cntCandidate += right[idx1] - left[idx1] + 1;

            ex = ex2 = 0;
            memset(T, 0, (size_t)2 * queryLength * sizeof(double));

            int begin1 = left[idx1] - shift - begin;
            int end1 = right[idx1] - shift + queryLength - 1 - begin;

            if (begin1 < 0) begin1 = 0;
            if (end1 > dataSize - 1) end1 = dataSize - 1;
            for (int i = begin1; i <= end1; i++) {
                double d = data[i];
                ex += d;
                ex2 += d * d;
                T[i % queryLength] = d;
                T[(i % queryLength) + queryLength] = d;

                if (i - begin1 >= queryLength - 1) {
                    int j = (i + 1) % queryLength;
                    windows++;

                    double mean = ex / queryLength;
                    double std = sqrt(ex2 / queryLength - mean * mean);

                    if (fabs(mean - meanQ) <= beta && (std / stdQ) <= alpha && (std / stdQ) >= 1.0 / alpha) {
                        passed++;
                        double dist = 0;
                        int kk;
                        for (kk = 0; kk < queryLength && (no_abandon || dist <= epsilon * epsilon); kk++) {
                            double x = (T[(order[kk] + j)] - mean) / std;
                            dist += (x - zQ[kk]) * (x - zQ[kk]);
                        }
                        terms += kk;
                        if (dist <= epsilon * epsilon) {
                            answers_add(&answers, begin + i - queryLength + 1, sqrt(dist));
                        }
                    }

                    ex -= T[j];
                    ex2 -= T[j] * T[j];
                }
            }
        }
    }
    free(zQ);
    free(order);
    free(Q_tmp);
    free(T);
    if (rc != KVO_OK) return rc;
    if (answers.overflow) return KVO_ERR_CAPACITY;
    stable_sort_answers(out_off, out_dist, answers.cnt);
    if (stats) {
        memset(stats, 0, KVO_NSTAT * sizeof(int64_t));
        stats[0] = cntCandidate;
        stats[1] = answers.cnt;
        stats[2] = windows;
        stats[3] = passed;
        stats[8] = terms;
    }
    return answers.cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* RSM-DTW phase 2 (K/QueryEngineDtw.java:349-452).  rho is the already-resolved band half-width */
/* (K/QueryEngineDtw.java:135-140 resolves r<=1 -> floor(r*m), else floor(r); see kvo_resolve_rho). */
int64_t kvo_query_rsm_dtw(const double *series, int n, const double *queryData, int length, const int32_t *left,
                          const int32_t *right, int k, int shift, double epsilon, int rho, int32_t *out_off,
                          double *out_dist, int64_t cap, int64_t *stats) {
    kvo_answers answers = {out_off, out_dist, cap, 0, 0};
    int64_t cntCandidate = 0, windows = 0, pKim = 0, pK1 = 0, pK2 = 0, cells = 0;

    double *Q = (double *)calloc((size_t)length, sizeof(double));
    for (int i = 0; i < length; i++) Q[i] = queryData[i];
    double *u = (double *)malloc((size_t)length * sizeof(double));
    double *l = (double *)malloc((size_t)length * sizeof(double));
    kvo_lower_upper_lemire(Q, length, rho, l, u);
    if (kvo_take_oob()) { /* K/QueryEngineDtw.java:362 dies in lowerUpperLemire: rho >= length */
        free(Q); free(u); free(l);
        return KVO_ERR_INDEX_OUT_OF_BOUNDS;
    }
    /* the reorder sort is commented out in the reference (:369-372): order = identity */
    int *order = (int *)malloc((size_t)length * sizeof(int));
    double *qo = (double *)malloc((size_t)length * sizeof(double));
    double *uo = (double *)malloc((size_t)length * sizeof(double));
    double *lo = (double *)malloc((size_t)length * sizeof(double));
    for (int i = 0; i < length; i++) {
        int o = i;
        order[i] = o;
        qo[i] = Q[o];
        uo[i] = u[o];
        lo[i] = l[o];
    }
    double *T = (double *)malloc((size_t)2 * length * sizeof(double));
    double *cb1 = (double *)malloc((size_t)length * sizeof(double));
    double *cb2 = (double *)malloc((size_t)length * sizeof(double));
    double *cT = (double *)malloc((size_t)length * sizeof(double));
    double *cb = (double *)malloc((size_t)length * sizeof(double));
    int rc = KVO_OK;

    for (int p = 0; p < k; p++) {
        cntCandidate += right[p] - left[p] + 1;

        int begin = left[p] - shift;
        int end = right[p] - shift + length - 1;
        if (begin < 1) begin = 1;
        if (end > n) end = n;
        const double *data;
        int dataSize = end - begin + 1;
        rc = read_time_series(series, n, begin, dataSize, &data);
        if (rc != KVO_OK) break;
        memset(T, 0, (size_t)2 * length * sizeof(double));

        double *uBuff = (double *)malloc((size_t)dataSize * sizeof(double));
        double *lBuff = (double *)malloc((size_t)dataSize * sizeof(double));
        kvo_lower_upper_lemire(data, dataSize, rho, lBuff, uBuff);
        if (kvo_take_oob()) { /* K/QueryEngineDtw.java:399: a fetched stub no longer than the band */
            free(uBuff); free(lBuff);
            rc = KVO_ERR_INDEX_OUT_OF_BOUNDS;
            break;
        }

        for (int i = 0; i < dataSize; i++) {
            double d = data[i];
            T[i % length] = d;
            T[(i % length) + length] = d;

            if (i >= length - 1) {
                int j = (i + 1) % length;
                windows++;

                double lbKim = kvo_lb_kim_hierarchy(T, Q, j, length, 0, 1, epsilon * epsilon);
                if (lbKim <= epsilon * epsilon) {
                    pKim++;
                    memset(cb1, 0, (size_t)length * sizeof(double)); /* new double[length] */
                    double lbK = kvo_lb_keogh_cumulative(order, T, uo, lo, cb1, j, length, 0, 1, epsilon * epsilon);
                    if (lbK <= epsilon * epsilon) {
                        pK1++;
                        memset(cb2, 0, (size_t)length * sizeof(double));
                        double lbK2 = kvo_lb_keogh_data_cumulative(order, qo, cb2, i - length + 1, lBuff, uBuff,
                                                                   length, 0, 1, epsilon * epsilon);
                        if (lbK2 <= epsilon * epsilon) {
                            pK2++;
                            memcpy(cT, T + j, (size_t)length * sizeof(double));
                            if (lbK > lbK2) {
                                cb[length - 1] = cb1[length - 1];
                                for (int kk = length - 2; kk >= 0; kk--) cb[kk] = cb[kk + 1] + cb1[kk];
                            } else {
                                cb[length - 1] = cb2[length - 1];
                                for (int kk = length - 2; kk >= 0; kk--) cb[kk] = cb[kk + 1] + cb2[kk];
                            }
                            double dist = kvo_dtw(cT, Q, cb, length, rho, epsilon * epsilon, &cells);
                            if (dist <= epsilon * epsilon) {
                                answers_add(&answers, begin + i - length + 1, sqrt(dist));
                            }
                        }
                    }
                }
            }
        }
        free(uBuff);
        free(lBuff);
    }
    free(Q); free(u); free(l); free(order); free(qo); free(uo); free(lo);
    free(T); free(cb1); free(cb2); free(cT); free(cb);
    if (rc != KVO_OK) return rc;
    if (answers.overflow) return KVO_ERR_CAPACITY;
    stable_sort_answers(out_off, out_dist, answers.cnt);
    if (stats) {
        memset(stats, 0, KVO_NSTAT * sizeof(int64_t));
        stats[0] = cntCandidate;
        stats[1] = answers.cnt;
        stats[2] = windows;
        stats[3] = windows;
        stats[4] = pKim;
        stats[5] = pK1;
        stats[6] = pK2;
        stats[7] = cells;
    }
    return answers.cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* cNSM-DTW phase 2 (K/NormQueryEngineDtw.java:205-211 statistics, :457-603 phase 2).           */
int64_t kvo_query_cnsm_dtw(const double *series, int n, const double *queryData, int queryLength,
                           const int32_t *left, const int32_t *right, int k, int shift, double epsilon, int rho,
                           double alpha, double beta, int32_t *out_off, double *out_dist, int64_t cap,
                           int64_t *stats) {
    kvo_answers answers = {out_off, out_dist, cap, 0, 0};
    int64_t cntCandidate = 0, windows = 0, passed = 0, pKim = 0, pK1 = 0, pK2 = 0, cells = 0;

    double ex = 0, ex2 = 0;
    for (int i = 0; i < queryLength; i++) {
        double value = queryData[i];
        ex += value;
        ex2 += value * value;
    }
    double meanQ = ex / queryLength;
    double stdQ = sqrt(ex2 / queryLength - meanQ * meanQ);

    double *zQ = (double *)calloc((size_t)queryLength, sizeof(double));
    for (int i = 0; i < queryLength; i++) zQ[i] = (queryData[i] - meanQ) / stdQ;
    double *u = (double *)malloc((size_t)queryLength * sizeof(double));
    double *l = (double *)malloc((size_t)queryLength * sizeof(double));
    kvo_lower_upper_lemire(zQ, queryLength, rho, l, u);
    if (kvo_take_oob()) { /* K/NormQueryEngineDtw.java:469, before the sort */
        free(zQ); free(u); free(l);
        return KVO_ERR_INDEX_OUT_OF_BOUNDS;
    }
    kvo_index *tmpQ = (kvo_index *)malloc((size_t)queryLength * sizeof(kvo_index));
    for (int i = 0; i < queryLength; i++) {
        tmpQ[i].value = zQ[i];
        tmpQ[i].index = i;
    }
    if (kvo_java_timsort(tmpQ, queryLength, cmp_abs_desc_trunc)) { /* Arrays.sort throws: query() dies with it (:475) */
        if (!kvo_lenient_sort) {
            free(zQ); free(u); free(l); free(tmpQ);
            return KVO_ERR_ILLEGAL_ARGUMENT;
        }
        for (int i = 0; i < queryLength; i++) {
            tmpQ[i].value = zQ[i];
            tmpQ[i].index = i;
        }
    }
    /* GUARD (not in the Java code): the chunk loop below indexes the fetched chunk with `left - begin`.  The engines only
     * ever pass the list sortAndMergeIntervals leaves (sorted by left, K/QueryEngine.java:664-693); on a list that is not
     * sorted by left Java ends in the range check of readTimeSeries or in an IndexOutOfBoundsException, depending on the
     * list, and C would read outside the chunk.  Reported as the IllegalArgument case (the library does the same) -- HERE,
     * where Java first touches the list: behind the query envelope (:469) and the sort (:475), whose own exceptions come
     * first (round 3: profiles/fuzz_lists.py `wide` found the guard in front of them). */
    for (int p = 1; p < k; p++)
        if (left[p] < left[p - 1]) {
            free(zQ); free(u); free(l); free(tmpQ);
            return KVO_ERR_ILLEGAL_ARGUMENT;
        }
    int *order = (int *)malloc((size_t)queryLength * sizeof(int));
    double *qo = (double *)malloc((size_t)queryLength * sizeof(double));
    double *uo = (double *)malloc((size_t)queryLength * sizeof(double));
    double *lo = (double *)malloc((size_t)queryLength * sizeof(double));
    for (int i = 0; i < queryLength; i++) {
        int o = tmpQ[i].index;
        order[i] = o;
        qo[i] = zQ[o];
        uo[i] = u[o];
        lo[i] = l[o];
    }
    double *T = (double *)malloc((size_t)2 * queryLength * sizeof(double));
    double *cb1 = (double *)malloc((size_t)queryLength * sizeof(double));
    double *cb2 = (double *)malloc((size_t)queryLength * sizeof(double));
    double *zT = (double *)malloc((size_t)queryLength * sizeof(double));
    double *cb = (double *)malloc((size_t)queryLength * sizeof(double));
    int rc = KVO_OK;

    int idx = 0;
    while (idx < k) {
        int beginIdx = idx, endIdx = idx;

        int begin = left[idx] - shift;
        int end = right[idx] - shift + queryLength - 1;
        if (begin < 1) begin = 1;
        int length = end - begin + 1;
        idx++;

        while (idx < k) {
            begin = left[idx] - shift;
            int newLength = length + begin - end - 1;
            end = right[idx] - shift + queryLength - 1;
            if (end > n) end = n;
            newLength += end - begin + 1;
            if (newLength > KVO_MAX_SCAN_DATA_LENGTH) break;
            endIdx = idx;
            length = newLength;
            idx++;
        }

        begin = left[beginIdx] - shift;
        end = right[endIdx] - shift + queryLength - 1;
        if (begin < 1) begin = 1;
        if (end > n) end = n;
        const double *data;
        int dataSize = end - begin + 1;
        rc = read_time_series(series, n, begin, dataSize, &data);
        if (rc != KVO_OK) break;

        double *uBuff = (double *)malloc((size_t)dataSize * sizeof(double));
        double *lBuff = (double *)malloc((size_t)dataSize * sizeof(double));
        kvo_lower_upper_lemire(data, dataSize, rho, lBuff, uBuff);
        if (kvo_take_oob()) { /* K/NormQueryEngineDtw.java:524 */
            free(uBuff); free(lBuff);
            rc = KVO_ERR_INDEX_OUT_OF_BOUNDS;
            break;
        }

        for (int idx1 = beginIdx; idx1 <= endIdx; idx1++) {
            cntCandidate += right[idx1] - left[idx1] + 1;

            ex = ex2 = 0;
            memset(T, 0, (size_t)2 * queryLength * sizeof(double));

            int begin1 = left[idx1] - shift - begin;
            int end1 = right[idx1] - shift + queryLength - 1 - begin;

            if (begin1 < 0) begin1 = 0;
            if (end1 > dataSize - 1) end1 = dataSize - 1;
            for (int i = begin1; i <= end1; i++) {
                double d = data[i];
                ex += d;
                ex2 += d * d;
                T[i % queryLength] = d;
                T[(i % queryLength) + queryLength] = d;

                if (i - begin1 >= queryLength - 1) {
                    int j = (i + 1) % queryLength;
                    windows++;

                    double mean = ex / queryLength;
                    double std = sqrt(ex2 / queryLength - mean * mean);

                    if (fabs(mean - meanQ) <= beta && (std / stdQ) <= alpha && (std / stdQ) >= 1.0 / alpha) {
                        passed++;
                        double lbKim = kvo_lb_kim_hierarchy(T, zQ, j, queryLength, mean, std, epsilon * epsilon);
                        if (lbKim <= epsilon * epsilon) {
                            pKim++;
                            memset(cb1, 0, (size_t)queryLength * sizeof(double));
                            double lbK = kvo_lb_keogh_cumulative(order, T, uo, lo, cb1, j, queryLength, mean, std,
                                                                 epsilon * epsilon);
                            if (lbK <= epsilon * epsilon) {
                                pK1++;
                                for (int kk = 0; kk < queryLength; kk++) {
                                    zT[kk] = (T[(kk + j)] - mean) / std;
                                }
                                memset(cb2, 0, (size_t)queryLength * sizeof(double));
                                double lbK2 = kvo_lb_keogh_data_cumulative(order, qo, cb2, i - queryLength + 1, lBuff,
                                                                           uBuff, queryLength, mean, std,
                                                                           epsilon * epsilon);
                                if (lbK2 <= epsilon * epsilon) {
                                    pK2++;
                                    if (lbK > lbK2) {
                                        cb[queryLength - 1] = cb1[queryLength - 1];
                                        for (int kk = queryLength - 2; kk >= 0; kk--) cb[kk] = cb[kk + 1] + cb1[kk];
                                    } else {
                                        cb[queryLength - 1] = cb2[queryLength - 1];
                                        for (int kk = queryLength - 2; kk >= 0; kk--) cb[kk] = cb[kk + 1] + cb2[kk];
                                    }
                                    double dist = kvo_dtw(zT, zQ, cb, queryLength, rho, epsilon * epsilon, &cells);
                                    if (dist <= epsilon * epsilon) {
                                        answers_add(&answers, begin + i - queryLength + 1, sqrt(dist));
                                    }
                                }
                            }
                        }
                    }

                    ex -= T[j];
                    ex2 -= T[j] * T[j];
                }
            }
        }
        free(uBuff);
        free(lBuff);
    }
    free(zQ); free(u); free(l); free(tmpQ); free(order); free(qo); free(uo); free(lo);
    free(T); free(cb1); free(cb2); free(zT); free(cb);
    if (rc != KVO_OK) return rc;
    if (answers.overflow) return KVO_ERR_CAPACITY;
    stable_sort_answers(out_off, out_dist, answers.cnt);
    if (stats) {
        memset(stats, 0, KVO_NSTAT * sizeof(int64_t));
        stats[0] = cntCandidate;
        stats[1] = answers.cnt;
        stats[2] = windows;
        stats[3] = passed;
        stats[4] = pKim;
        stats[5] = pK1;
        stats[6] = pK2;
        stats[7] = cells;
    }
    return answers.cnt;
}

/* rho resolution used by every DTW entry point: `r <= 1 -> floor(r*m) else floor(r)`
 * (K/QueryEngineDtw.java:135-140; K/NormQueryEngineDtw.java has the same prompt logic). */
int kvo_resolve_rho(double r, int m) {
    if (r <= 1) return (int)floor(r * m);
    return (int)floor(r);
}

/* Query statistics as the cNSM engines compute them (K/NormQueryEngine.java:192-198). */
void kvo_query_stats(const double *queryData, int m, double *meanQ, double *stdQ) {
    double ex = 0, ex2 = 0;
    for (int i = 0; i < m; i++) {
        ex += queryData[i];
        ex2 += queryData[i] * queryData[i];
    }
    *meanQ = ex / m;
    *stdQ = sqrt(ex2 / m - (*meanQ) * (*meanQ));
}

/* Full (unpruned, no early abandon) banded DTW through kvo_dtw: cb = 0, bsf = +inf-ish so the
 * abandon test never fires.  Used by tests to get the exact DP value of any candidate. */
double kvo_dtw_full(const double *A, const double *B, int m, int r) {
    double *cb = (double *)calloc((size_t)m, sizeof(double));
    double v = kvo_dtw(A, B, cb, m, r, 1e300, NULL);
    free(cb);
    return v;
}

/* ================================================================================================================
 * Index build, step 1 ("fixed-width index rows"), K/IndexBuilder.java:186-300 for one window length w, on a data file
 * of exactly n points (K/operator/file/TimeSeriesFileOperator.java readAllTimeSeries).  Test infrastructure, like the
 * rest of this file (the GPU side is kvm_index_rows).
 *
 * The builder streams the series in buffers of EPOCH = 100000 points (:136); every buffer after the first starts with
 * the last w-1 points of the previous one (:203-208), and the running sums ex / ex2 restart at every buffer (:232-233)
 * -- "for reducing the floating point error" (:135).  For every window position loc (1-based start offset, :257)
 *     mean = ex / w (:247),  key = MeanIntervalUtils.toRound(mean) (:263, K/utils/MeanIntervalUtils.java:50-60)
 * and consecutive positions with the same key (Double.equals, i.e. bit pattern) extend the current interval of that
 * key's row until it spans MAXIMUM_DIFF - 1 = 255 positions (:266, K/common/entity/IndexNode.java:31); otherwise a new
 * interval [loc, loc] is started.  Output: the intervals in position order, each with its key.  (Grouping them into
 * rows by key -- the HashMap of :192,:270-276 -- and step 2's row merging are host logic on top of this list.)
 * ================================================================================================================ */
double kvo_to_round(double value) { /* K/utils/MeanIntervalUtils.java:50-60, posOfD = 2 */
    value *= 10.0;                  /* Math.pow(10, posOfD - 1) == 10.0 */
    double intValue = floor(value);
    double diff = value - intValue;
    double retValue = intValue;
    if (double_compare(diff, 0.5) >= 0) retValue += 0.5;
    retValue *= 0.1;                /* Math.pow(10, -posOfD + 1) == 0.1 (the double nearest to 1/10) */
    return retValue;
}

static int double_equals(double a, double b) { /* Double.equals: doubleToLongBits equality (NaN == NaN, 0.0 != -0.0) */
    if (a != a && b != b) return 1;
    uint64_t ua, ub;
    memcpy(&ua, &a, 8);
    memcpy(&ub, &b, 8);
    return ua == ub;
}

/* Returns the number of intervals (also when it exceeds cap: only the first cap are written); mean_out (optional,
 * n - w + 1 doubles) receives the reference's mean of every window position. */
int64_t kvo_index_rows(const double *series, int64_t n, int32_t w, double *keys, int32_t *first, int32_t *last,
                       int64_t cap, double *mean_out) {
    const int EPOCH = 100000;                                            /* :136 */
    if (w < 1 || w > EPOCH || n < w) return 0;
    double *buffer = (double *)malloc(sizeof(double) * EPOCH);           /* :147 */
    double *t = (double *)malloc(sizeof(double) * 2 * (size_t)w);        /* :146 */
    int64_t cnt_rows = 0, next = 0;                                      /* next: index of the next unread point */
    int has_last = 0;
    double lastMeanRound = 0.0;
    int64_t cur_first = 0;
    int done = 0;
    int64_t it = 0;
    while (!done) {                                                      /* :195 */
        int ep;
        if (it == 0) {                                                   /* :197-204 */
            for (int k = 0; k < w - 1; k++)
                if (next < n) buffer[k] = series[next++];
        } else {
            for (int k = 0; k < w - 1; k++) buffer[k] = buffer[EPOCH - w + 1 + k]; /* :205-208 */
        }
        ep = w - 1;                                                      /* :211 */
        while (ep < EPOCH) {                                             /* :212-220 */
            if (next < n) buffer[ep++] = series[next++];
            else break;
        }
        if (ep <= w - 1) {                                               /* :224-225 */
            done = 1;
        } else {
            double ex = 0, ex2 = 0;                                      /* :232-233 */
            for (int i = 0; i < ep; i++) {                               /* :234 */
                double d = buffer[i];
                ex += d;                                                 /* :239 */
                ex2 += d * d;                                            /* :240 */
                t[i % w] = d;                                            /* :243 */
                t[(i % w) + w] = d;                                      /* :246 */
                if (i >= w - 1) {                                        /* :249 */
                    double mean = ex / w;                                /* :250 */
                    int j = (i + 1) % w;                                 /* :255 */
                    int64_t loc = it * (int64_t)(EPOCH - w + 1) + i - w + 1 + 1; /* :258 */
                    if (loc > n) {                                       /* :259-262 */
                        done = 1;
                        break;
                    }
                    double curMeanRound = kvo_to_round(mean);            /* :264 */
                    if (mean_out) mean_out[loc - 1] = mean;
                    if (!has_last || !double_equals(lastMeanRound, curMeanRound) ||
                        loc - cur_first == 256 - 1) {                    /* :267 (IndexNode.MAXIMUM_DIFF - 1) */
                        if (cnt_rows < cap) {                            /* new interval [loc, loc], :277 */
                            keys[cnt_rows] = curMeanRound;
                            if (curMeanRound != curMeanRound) { /* one NaN key (Double.equals / doubleToLongBits) */
                                const uint64_t canon = 0x7ff8000000000000ULL;
                                memcpy(&keys[cnt_rows], &canon, 8);
                            }
                            first[cnt_rows] = (int32_t)loc;
                            last[cnt_rows] = (int32_t)loc;
                        }
                        cnt_rows++;
                        cur_first = loc;
                        lastMeanRound = curMeanRound;
                        has_last = 1;
                    } else if (cnt_rows - 1 < cap) {
                        last[cnt_rows - 1] = (int32_t)loc;               /* :283 */
                    }
                    ex -= t[j];                                          /* :287 */
                    ex2 -= t[j] * t[j];                                  /* :288 */
                }
            }
            if (ep < EPOCH) done = 1;                                    /* :293-294 */
            else it++;
        }
    }
    free(buffer);
    free(t);
    return cnt_rows;
}

/* ================================================================================================================
 * Index build, the rest of IndexBuilder.run (K/IndexBuilder.java:296-347) and the index file
 * (K/operator/file/IndexFileOperator.java:127-164, K/common/entity/IndexNode.java:51-95, K/utils/ByteUtils.java,
 * K/utils/IndexNodeUtils.java:30-90): rows = one IndexNode per distinct key (the HashMap of :192,:270-284), step 2
 * merges adjacent rows, the file holds the merged rows in ascending key order (TreeMap), the statistic triples and the
 * offset table.  Everything below is int / byte work on top of kvo_index_rows.
 * ================================================================================================================ */
typedef struct {
    int32_t *first, *last; /* positions: List<Pair<Integer, Integer>> */
    int64_t size, cap;
} kvo_node;

static void node_push(kvo_node *nd, int32_t f, int32_t l) {
    if (nd->size == nd->cap) {
        nd->cap = nd->cap ? nd->cap * 2 : 16;
        nd->first = (int32_t *)realloc(nd->first, (size_t)nd->cap * sizeof(int32_t));
        nd->last = (int32_t *)realloc(nd->last, (size_t)nd->cap * sizeof(int32_t));
    }
    nd->first[nd->size] = f;
    nd->last[nd->size] = l;
    nd->size++;
}

static void node_free(kvo_node *nd) {
    free(nd->first);
    free(nd->last);
    memset(nd, 0, sizeof *nd);
}

/* IndexNodeUtils.addInterval (K/utils/IndexNodeUtils.java:82-90): "diff can not exceed 255" */
static void node_add_interval(kvo_node *nd, int32_t f, int32_t l) {
    while (l - f >= 256) { /* IndexNode.MAXIMUM_DIFF */
        int32_t newFirst = f + 256 - 1;
        node_push(nd, f, newFirst);
        f = newFirst + 1;
    }
    node_push(nd, f, l);
}

/* IndexNodeUtils.mergeIndexNode (K/utils/IndexNodeUtils.java:30-80) */
static void node_merge(const kvo_node *node1, const kvo_node *node2, kvo_node *ret) {
    int64_t index1 = 0, index2 = 0;
    int has1 = 0, has2 = 0;
    int32_t l1f = 0, l1s = 0, l2f = 0, l2s = 0; /* last1, last2: working copies */
    while (index1 < node1->size && index2 < node2->size) {
        if (!has1) { l1f = node1->first[index1]; l1s = node1->last[index1]; has1 = 1; }
        if (!has2) { l2f = node2->first[index2]; l2s = node2->last[index2]; has2 = 1; }
        if (l1s + 1 < l2f) {
            node_add_interval(ret, l1f, l1s);
            index1++;
            has1 = 0;
        } else if (l2s + 1 < l1f) {
            node_add_interval(ret, l2f, l2s);
            index2++;
            has2 = 0;
        } else {
            if (l1s < l2s) {
                if (l1f < l2f) l2f = l1f;
                index1++;
                has1 = 0;
            } else {
                if (l2f < l1f) l1f = l2f;
                index2++;
                has2 = 0;
            }
        }
    }
    for (int64_t i = index1; i < node1->size; i++) {
        if (!has1) { l1f = node1->first[i]; l1s = node1->last[i]; }
        node_add_interval(ret, l1f, l1s);
        has1 = 0;
    }
    for (int64_t i = index2; i < node2->size; i++) {
        if (!has2) { l2f = node2->first[i]; l2s = node2->last[i]; }
        node_add_interval(ret, l2f, l2s);
        has2 = 0;
    }
}

static void put_be32(uint8_t *p, int32_t v) { /* Bytes.toBytes(int): big-endian */
    uint32_t u = (uint32_t)v;
    p[0] = (uint8_t)(u >> 24); p[1] = (uint8_t)(u >> 16); p[2] = (uint8_t)(u >> 8); p[3] = (uint8_t)u;
}

static void put_be_double(uint8_t *p, double v) { /* Bytes.toBytes(double): doubleToRawLongBits, big-endian */
    uint64_t u;
    memcpy(&u, &v, 8);
    for (int b = 0; b < 8; b++) p[b] = (uint8_t)(u >> (56 - 8 * b));
}

/* IndexNode.toBytesCompact (K/common/entity/IndexNode.java:61-100); `out` has room for 8 * size bytes; returns length */
static int64_t node_to_bytes_compact(const kvo_node *nd, uint8_t *result) {
    int64_t index = 0, length = 0;
    int count = 0, isPacking = 0;
    while (index < nd->size) {
        if (!isPacking) {
            put_be32(result + length, nd->first[index]);
            length += 4 + 1; /* first: 4 bytes, remain 1 byte for count */
            int diff = nd->last[index] - nd->first[index];
            result[length++] = (uint8_t)(int8_t)(diff - 128);
            isPacking = 1;
            count = 1;
        } else {
            int diff = nd->first[index] - nd->last[index - 1];
            if (diff < 256 && (count - 1) / 2 + 2 < 256) {
                result[length++] = (uint8_t)(int8_t)(diff - 128);
                diff = nd->last[index] - nd->first[index];
                result[length++] = (uint8_t)(int8_t)(diff - 128);
                count += 2;
            } else {
                result[length - count - 1] = (uint8_t)(int8_t)((count - 1) / 2 - 128);
                isPacking = 0;
                continue;
            }
        }
        index++;
    }
    if (isPacking) result[length - count - 1] = (uint8_t)(int8_t)((count - 1) / 2 - 128);
    return length;
}

typedef struct {
    uint64_t asc; /* Double.compareTo order as an unsigned key */
    int64_t seq;
} kvo_keyed;

static uint64_t compare_to_bits(double v) { /* monotone in Double.compareTo: -0.0 < 0.0, NaN (canonical) above +Inf */
    uint64_t u;
    if (v != v) u = 0x7ff8000000000000ULL;
    else memcpy(&u, &v, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}

static int keyed_cmp(const void *a, const void *b) {
    const kvo_keyed *x = (const kvo_keyed *)a, *y = (const kvo_keyed *)b;
    if (x->asc != y->asc) return x->asc < y->asc ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}

/* IndexBuilder.run for one window length `w` (Wu), whole: step 1 (kvo_index_rows), the rows, step 2, and the bytes
 * IndexFileOperator.writeAll puts into files/index-{N}-{Wu}.  Returns the file length (also when > cap: then nothing is
 * written), or KVO_ERR_ILLEGAL_ARGUMENT when there is no window at all (the Java code runs into
 * rawStatisticInfo.get(0) on an empty list, :321).  info (optional, 4 int64): rows before step 2, rows after, intervals
 * before, intervals after. */
int64_t kvo_index_file(const double *series, int64_t n, int32_t w, uint8_t *out, int64_t cap, int64_t *info) {
    int64_t pos_cap = n - w + 1;
    if (w < 1 || w > 100000 || pos_cap < 1) return KVO_ERR_ILLEGAL_ARGUMENT;
    double *keys = (double *)malloc((size_t)pos_cap * sizeof(double));
    int32_t *first = (int32_t *)malloc((size_t)pos_cap * sizeof(int32_t));
    int32_t *last = (int32_t *)malloc((size_t)pos_cap * sizeof(int32_t));
    const int64_t cnt = kvo_index_rows(series, n, w, keys, first, last, pos_cap, NULL);
    /* the HashMap<Double, IndexNode> of :192: one node per distinct key, intervals in the order they were appended */
    kvo_keyed *kd = (kvo_keyed *)malloc((size_t)cnt * sizeof(kvo_keyed));
    for (int64_t i = 0; i < cnt; i++) {
        kd[i].asc = compare_to_bits(keys[i]);
        kd[i].seq = i;
    }
    qsort(kd, (size_t)cnt, sizeof(kvo_keyed), keyed_cmp);
    int64_t nrows = 0;
    for (int64_t i = 0; i < cnt; i++) nrows += (i == 0 || kd[i].asc != kd[i - 1].asc);
    kvo_node *rows = (kvo_node *)calloc((size_t)nrows, sizeof(kvo_node)); /* ascending key */
    double *row_key = (double *)malloc((size_t)nrows * sizeof(double));
    {
        int64_t r = -1;
        for (int64_t i = 0; i < cnt; i++) {
            if (i == 0 || kd[i].asc != kd[i - 1].asc) row_key[++r] = keys[kd[i].seq];
            node_push(&rows[r], first[kd[i].seq], last[kd[i].seq]);
        }
    }
    free(kd); free(keys); free(first); free(last);
    /* step 2 (:308-347).  rawStatisticInfo sorted by DESCENDING key (:318) = rows[nrows-1 .. 0]; StatisticInfo.append /
     * getAverage (K/statistic/StatisticInfo.java:43-63): a double sum of the sizes over n */
    double sum = 0;
    for (int64_t r = 0; r < nrows; r++) sum += (double)rows[r].size;
    const double average = sum / (double)nrows;
    /* indexStore (TreeMap, ascending) + statisticInfo (later sorted ascending, stable, keys distinct): collect as
     * produced (descending), reverse at the end */
    kvo_node *store = (kvo_node *)calloc((size_t)nrows, sizeof(kvo_node));
    double *store_key = (double *)malloc((size_t)nrows * sizeof(double));
    int64_t nstore = 0, iv_before = cnt, iv_after = 0;
    kvo_node lastn = rows[nrows - 1]; /* indexNodeMap.get(rawStatisticInfo.get(0).getFirst()) */
    int last_owned = 0;               /* whether lastn is a merged node this function allocated */
    for (int64_t i = 1; i < nrows; i++) {
        const kvo_node *current = &rows[nrows - 1 - i];
        int isMerged = 0;
        if ((double)current->size < average * 1.2) { /* rawStatisticInfo.get(i).getSecond().getFirst() < average.getAverage() * 1.2 */
            kvo_node merged = {0};
            node_merge(&lastn, current, &merged);
            if ((double)merged.size < (double)(lastn.size + current->size) * 0.8) {
                if (last_owned) node_free(&lastn);
                lastn = merged;
                last_owned = 1;
                isMerged = 1;
            } else {
                node_free(&merged);
            }
        }
        if (!isMerged) {
            store_key[nstore] = row_key[nrows - 1 - (i - 1)]; /* rawStatisticInfo.get(i - 1).getFirst() */
            if (last_owned) {
                store[nstore] = lastn;
            } else { /* copy: the rows array is freed below */
                for (int64_t t = 0; t < lastn.size; t++) node_push(&store[nstore], lastn.first[t], lastn.last[t]);
            }
            nstore++;
            lastn = *current;
            last_owned = 0;
        }
    }
    store_key[nstore] = row_key[0]; /* rawStatisticInfo.get(size - 1).getFirst() */
    if (last_owned) {
        store[nstore] = lastn;
    } else {
        for (int64_t t = 0; t < lastn.size; t++) node_push(&store[nstore], lastn.first[t], lastn.last[t]);
    }
    nstore++;
    for (int64_t r = 0; r < nrows; r++) node_free(&rows[r]);
    free(rows); free(row_key);
    /* file image: rows ascending (store[] is descending), then the triples, then the offsets */
    int64_t body = 0;
    for (int64_t r = 0; r < nstore; r++) {
        body += 8 + 8 * store[r].size; /* upper bound of a line */
        iv_after += store[r].size;
    }
    uint8_t *img = (uint8_t *)malloc((size_t)(body + 16 * nstore + 4 * (nstore + 2) + 64));
    int32_t *offsets = (int32_t *)malloc((size_t)(nstore + 2) * sizeof(int32_t));
    int64_t offset = 0, noff = 0;
    for (int64_t r = nstore - 1; r >= 0; r--) { /* writeIndex (:142-150) */
        offsets[noff++] = (int32_t)offset;
        put_be_double(img + offset, store_key[r]);
        offset += 8;
        offset += node_to_bytes_compact(&store[r], img + offset);
    }
    /* writeStatisticInfo (:152-160) + ByteUtils.listTripleToByteArray: running (int) sums of #intervals and #offsets */
    offsets[noff++] = (int32_t)offset;
    int32_t cum_iv = 0, cum_off = 0;
    for (int64_t r = nstore - 1; r >= 0; r--) {
        int32_t nofs = 0;
        for (int64_t t = 0; t < store[r].size; t++) nofs += store[r].last[t] - store[r].first[t] + 1; /* getNumOfOffsets */
        cum_iv = (int32_t)((uint32_t)cum_iv + (uint32_t)store[r].size);
        cum_off = (int32_t)((uint32_t)cum_off + (uint32_t)nofs);
        put_be_double(img + offset, store_key[r]);
        put_be32(img + offset + 8, cum_iv);
        put_be32(img + offset + 12, cum_off);
        offset += 16;
    }
    offsets[noff++] = (int32_t)offset; /* writeOffsetInfo (:162-165) */
    for (int64_t t = 0; t < noff; t++) {
        put_be32(img + offset, offsets[t]);
        offset += 4;
    }
    if (info) {
        info[0] = nrows;
        info[1] = nstore;
        info[2] = iv_before;
        info[3] = iv_after;
    }
    if (offset <= cap && out) memcpy(out, img, (size_t)offset);
    for (int64_t r = 0; r < nstore; r++) node_free(&store[r]);
    free(store); free(store_key); free(img); free(offsets);
    return offset;
}

/* ------------------------------------------------------------------------------------------ */
/* Index-free full scans: the UCR-suite style executors the reference's experiments use as ground truth
 * (K/experiments/NormQueryTestGroupBySelectivity.java:115-126, NormQueryDtwTestGroupBySelectivity.java:116).
 * `U/` = src/main/java/cn/edu/fudan/dsm/kvmatch/experiments/ucr/.  The scanner (readAllTimeSeries) is restated as a
 * pass over the first n points of the series; N == n (the reference's `++cnt <= N` guard, U/UcrEdQueryExecutor.java:60-63,
 * skips the count at node boundaries and is only a loose upper limit). */

/* U/UcrEdQueryExecutor.java:96-175.  One pair of running sums over the WHOLE series (never restarted); answers are
 * 1-based offsets (`i - M + 2`, :161).  The inner loop abandons on `sum < bsf` (:88), acceptance is `dist <= eps^2`. */
int64_t kvo_ucr_ed(const double *series, int n, const double *queryData, int M, double Epsilon, double Alpha, double Beta,
                   int32_t *out_off, double *out_dist, int64_t cap, int64_t *stats) {
    kvo_answers answers = {out_off, out_dist, cap, 0, 0};
    int64_t windows = 0, passed = 0;
    double ex = 0, ex2 = 0;
    double *Q = (double *)malloc((size_t)M * sizeof(double));
    for (int i = 0; i < M; i++) { /* :98-103 */
        double d = queryData[i];
        ex += d;
        ex2 += d * d;
        Q[i] = d;
    }
    double meanQ = ex / M; /* :106-108 */
    double stdQ = ex2 / M;
    stdQ = sqrt(stdQ - meanQ * meanQ);
    for (int i = 0; i < M; i++) Q[i] = (Q[i] - meanQ) / stdQ;
    int *order = (int *)malloc((size_t)M * sizeof(int));
    kvo_index *Q_tmp = (kvo_index *)malloc((size_t)M * sizeof(kvo_index));
    for (int i = 0; i < M; i++) {
        Q_tmp[i].value = Q[i];
        Q_tmp[i].index = i;
    }
    stable_sort_index(Q_tmp, M, cmp_abs_desc); /* :119-123, Double.compare(|o2|, |o1|) */
    for (int i = 0; i < M; i++) {
        Q[i] = Q_tmp[i].value;
        order[i] = Q_tmp[i].index;
    }
    double *t = (double *)calloc((size_t)2 * M, sizeof(double));
    const double bsf = Epsilon * Epsilon;
    ex = ex2 = 0;
    for (int i = 0; i < n; i++) { /* :143-171 */
        double d = series[i];
        ex += d;
        ex2 += d * d;
        t[i % M] = d;
        t[(i % M) + M] = d;
        if (i >= M - 1) {
            int j = (i + 1) % M;
            windows++;
            double mean = ex / M;
            double std = ex2 / M;
            std = sqrt(std - mean * mean);
            if (fabs(mean - meanQ) <= Beta && (std / stdQ) <= Alpha && (std / stdQ) >= 1.0 / Alpha) {
                passed++;
                double sum = 0; /* distance(), :85-92 */
                for (int k = 0; k < M && sum < bsf; k++) {
                    double x = (t[(order[k] + j)] - mean) / std;
                    sum += (x - Q[k]) * (x - Q[k]);
                }
                if (sum <= Epsilon * Epsilon) answers_add(&answers, i - M + 2, sqrt(sum));
            }
            ex -= t[j];
            ex2 -= t[j] * t[j];
        }
    }
    free(Q); free(order); free(Q_tmp); free(t);
    if (answers.overflow) return KVO_ERR_CAPACITY;
    stable_sort_answers(out_off, out_dist, answers.cnt);
    if (stats) {
        memset(stats, 0, KVO_NSTAT * sizeof(int64_t));
        stats[0] = windows;
        stats[1] = answers.cnt;
        stats[2] = windows;
        stats[3] = passed;
    }
    return answers.cnt;
}

/* U/UcrDtwQueryExecutor.java:84-315.  Buffers of EPOCH = 100000 points carrying M - 1 (:97,168-197); running sums
 * and the data envelope restart per buffer (:204,213-214); answers are 0-BASED offsets
 * (`it * (EPOCH - M + 1) + i - M + 1`, :283).  cb1/cb2 are reused without clearing (:157-161); a stale entry is only
 * ever read after a bound has already reached bsf.  The query order uses the truncating comparator of
 * K/NormQueryEngineDtw.java:475-478 (U/UcrDtwQueryExecutor.java:142-145): kvo_java_timsort, as in kvo_query_cnsm_dtw. */
int64_t kvo_ucr_dtw(const double *series, int n, const double *queryData, int M, double Epsilon, int Rho, double Alpha,
                    double Beta, int32_t *out_off, double *out_dist, int64_t cap, int64_t *stats) {
    const int EPOCH = 100000;
    if (M > EPOCH || M < 1) return KVO_ERR_ILLEGAL_ARGUMENT;
    kvo_answers answers = {out_off, out_dist, cap, 0, 0};
    int64_t windows = 0, passed = 0, pKim = 0, pK1 = 0, pK2 = 0, cells = 0;
    double *q = (double *)malloc((size_t)M * sizeof(double));
    double *qo = (double *)malloc((size_t)M * sizeof(double));
    double *uo = (double *)malloc((size_t)M * sizeof(double));
    double *lo = (double *)malloc((size_t)M * sizeof(double));
    int *order = (int *)malloc((size_t)M * sizeof(int));
    kvo_index *Q_tmp = (kvo_index *)malloc((size_t)M * sizeof(kvo_index));
    double *u = (double *)malloc((size_t)M * sizeof(double));
    double *l = (double *)malloc((size_t)M * sizeof(double));
    double *cb = (double *)calloc((size_t)M, sizeof(double));
    double *cb1 = (double *)calloc((size_t)M, sizeof(double));
    double *cb2 = (double *)calloc((size_t)M, sizeof(double));
    double *t = (double *)calloc((size_t)2 * M, sizeof(double));
    double *tz = (double *)malloc((size_t)M * sizeof(double));
    double *buffer = (double *)calloc((size_t)EPOCH, sizeof(double));
    double *u_buff = (double *)malloc((size_t)EPOCH * sizeof(double));
    double *l_buff = (double *)malloc((size_t)EPOCH * sizeof(double));
    double ex = 0, ex2 = 0;
    for (int i = 0; i < M; i++) { /* :118-123 */
        double d = queryData[i];
        ex += d;
        ex2 += d * d;
        q[i] = d;
    }
    double meanQ = ex / M; /* :126-131 */
    double stdQ = ex2 / M;
    stdQ = sqrt(stdQ - meanQ * meanQ);
    for (int i = 0; i < M; i++) q[i] = (q[i] - meanQ) / stdQ;
    kvo_lower_upper_lemire(q, M, Rho, l, u); /* :134 */
    if (kvo_take_oob()) { /* Rho >= M: ArrayIndexOutOfBoundsException */
        free(q); free(qo); free(uo); free(lo); free(order); free(Q_tmp); free(u); free(l);
        free(cb); free(cb1); free(cb2); free(t); free(tz); free(buffer); free(u_buff); free(l_buff);
        return KVO_ERR_INDEX_OUT_OF_BOUNDS;
    }
    for (int i = 0; i < M; i++) {
        Q_tmp[i].value = q[i];
        Q_tmp[i].index = i;
    }
    if (kvo_java_timsort(Q_tmp, M, cmp_abs_desc_trunc)) { /* :142-145; Arrays.sort throws and run() dies with it */
        if (!kvo_lenient_sort) {
            free(q); free(qo); free(uo); free(lo); free(order); free(Q_tmp); free(u); free(l);
            free(cb); free(cb1); free(cb2); free(t); free(tz); free(buffer); free(u_buff); free(l_buff);
            return KVO_ERR_ILLEGAL_ARGUMENT;
        }
        for (int i = 0; i < M; i++) {
            Q_tmp[i].value = q[i];
            Q_tmp[i].index = i;
        }
    }
    for (int i = 0; i < M; i++) { /* :146-154 */
        int o = Q_tmp[i].index;
        order[i] = o;
        qo[i] = q[o];
        uo[i] = u[o];
        lo[i] = l[o];
    }
    const double bsf = Epsilon * Epsilon;
    int64_t next = 0; /* scanner position */
    int done = 0, it = 0;
    while (!done) {
        if (it == 0) { /* :170-180 */
            for (int k = 0; k < M - 1; k++)
                if (next < n) buffer[k] = series[next++];
        } else { /* :181-185 */
            for (int k = 0; k < M - 1; k++) buffer[k] = buffer[EPOCH - M + 1 + k];
        }
        int ep = M - 1; /* :188-198 */
        while (ep < EPOCH) {
            if (next >= n) break;
            buffer[ep] = series[next++];
            ep++;
        }
        if (ep <= M - 1) { /* :202 */
            done = 1;
        } else {
            kvo_lower_upper_lemire(buffer, ep, Rho, l_buff, u_buff); /* :205 */
            ex = 0;
            ex2 = 0;
            for (int i = 0; i < ep; i++) { /* :215-296 */
                double d = buffer[i];
                ex += d;
                ex2 += d * d;
                t[i % M] = d;
                t[(i % M) + M] = d;
                if (i >= M - 1) {
                    windows++;
                    double mean = ex / M;
                    double std = ex2 / M;
                    std = sqrt(std - mean * mean);
                    int j = (i + 1) % M;
                    int I = i - (M - 1);
                    if (fabs(mean - meanQ) <= Beta && (std / stdQ) <= Alpha && (std / stdQ) >= 1.0 / Alpha) {
                        passed++;
                        double lb_kim = kvo_lb_kim_hierarchy(t, q, j, M, mean, std, bsf);
                        if (lb_kim <= Epsilon * Epsilon) {
                            pKim++;
                            double lb_k = kvo_lb_keogh_cumulative(order, t, uo, lo, cb1, j, M, mean, std, bsf);
                            if (lb_k <= Epsilon * Epsilon) {
                                pK1++;
                                for (int k = 0; k < M; k++) tz[k] = (t[(k + j)] - mean) / std;
                                double lb_k2 = kvo_lb_keogh_data_cumulative(order, qo, cb2, I, l_buff, u_buff, M, mean, std, bsf);
                                if (lb_k2 <= Epsilon * Epsilon) {
                                    pK2++;
                                    if (lb_k > lb_k2) {
                                        cb[M - 1] = cb1[M - 1];
                                        for (int k = M - 2; k >= 0; k--) cb[k] = cb[k + 1] + cb1[k];
                                    } else {
                                        cb[M - 1] = cb2[M - 1];
                                        for (int k = M - 2; k >= 0; k--) cb[k] = cb[k + 1] + cb2[k];
                                    }
                                    double dist = kvo_dtw(tz, q, cb, M, Rho, bsf, &cells);
                                    if (dist <= Epsilon * Epsilon)
                                        answers_add(&answers, (int32_t)((int64_t)it * (EPOCH - M + 1) + i - M + 1), sqrt(dist));
                                }
                            }
                        }
                    }
                    ex -= t[j];
                    ex2 -= t[j] * t[j];
                }
            }
            if (ep < EPOCH) done = 1; /* :299-303 */
            else it++;
        }
    }
    free(q); free(qo); free(uo); free(lo); free(order); free(Q_tmp); free(u); free(l);
    free(cb); free(cb1); free(cb2); free(t); free(tz); free(buffer); free(u_buff); free(l_buff);
    if (answers.overflow) return KVO_ERR_CAPACITY;
    stable_sort_answers(out_off, out_dist, answers.cnt);
    if (stats) {
        memset(stats, 0, KVO_NSTAT * sizeof(int64_t));
        stats[0] = windows;
        stats[1] = answers.cnt;
        stats[2] = windows;
        stats[3] = passed;
        stats[4] = pKim;
        stats[5] = pK1;
        stats[6] = pK2;
        stats[7] = cells;
    }
    return answers.cnt;
}
