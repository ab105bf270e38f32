"""An INDEPENDENT pure-Python transliteration of the reference's four phase-2 blocks, written from the Java sources (not
from oracle/kvm_oracle.c), statement for statement, to lower the risk that the C oracle mis-transcribes them:

    rsm_ed    K/QueryEngine.java:341-363, sort :373
    cnsm_ed   K/NormQueryEngine.java:192-198 (query statistics), :437-526, sort
    rsm_dtw   K/QueryEngineDtw.java:352-449, sort
    cnsm_dtw  K/NormQueryEngineDtw.java:205-211, :460-601, sort
    ucr_ed    K/experiments/ucr/UcrEdQueryExecutor.java:85-175;  ucr_dtw  K/experiments/ucr/UcrDtwQueryExecutor.java:84-315
    DtwUtils  K/utils/DtwUtils.java:24-337, CircularArray K/common/CircularArray.java:21-110,
    readTimeSeries range check K/operator/file/TimeSeriesFileOperator.java:55-57

(K/ = /root/reference/src/main/java/cn/edu/fudan/dsm/kvmatch/.)  Python floats are IEEE-754 binary64, `a * b + c` is
never fused and math.sqrt is correctly rounded, exactly like Java's double arithmetic; ints do not overflow at these
sizes.  TEST INFRASTRUCTURE ONLY; it pins nothing by itself (no output of the Java code exists to compare with), it is a
second, independently written reading of the same text, compared bit for bit with the C oracle in tests/test_oracle.py.

The one place where Java's behaviour is not a function of the text alone: NormQueryEngineDtw sorts with the comparator
(int)(|o2| - |o1|) (:475-478), which violates the Comparator contract, so the permutation depends on java.util.TimSort's
merge pattern.  java_arrays_sort at the end of this file restates JDK 8's TimSort for that call; where it throws
("Comparison method violates its general contract!") cnsm_dtw raises IllegalArgumentException like query() does."""
import functools
import math

INF = 1e20                      # DtwUtils.INF
WU0 = 25                        # WuList[0]
MAX_SCAN_DATA_LENGTH = 40000    # NormQueryEngine.MAX_SCAN_DATA_LENGTH / NormQueryEngineDtw


class ArrayIndexOutOfBoundsException(Exception):
    """java.lang.ArrayIndexOutOfBoundsException: the reference does not catch it -- query() ends with it.  Raised where Python's
    negative indices would silently wrap (DtwUtils.lowerUpperLemire on an array no longer than the band: `u[i - r - 1]`)."""


class IllegalArgumentException(Exception):
    pass


def read_time_series(series, left, length):
    """TimeSeriesFileOperator.readTimeSeries: 1-based, throws on a bad range."""
    n = len(series)
    if left < 1 or left + length - 1 > n or length < 1:
        raise IllegalArgumentException(f"left: {left}, length: {length}")
    return series[left - 1: left - 1 + length]


def _java_double_compare(a, b):
    """Double.compare: -0.0 < 0.0, NaN greater than everything and equal to itself."""
    if a < b:
        return -1
    if a > b:
        return 1
    ba, bb = math.copysign(1.0, a) if a == a else 2.0, math.copysign(1.0, b) if b == b else 2.0
    if a == b and a == 0.0:
        return (ba > bb) - (ba < bb)
    if a == b:
        return 0
    return (ba > bb) - (ba < bb)  # at least one NaN


def _sorted_answers(answers):
    # answers.sort(Comparator.comparing(Pair::getSecond)): List.sort is stable
    return sorted(answers, key=functools.cmp_to_key(lambda p1, p2: _java_double_compare(p1[1], p2[1])))


# ---------------------------------------------------------------------------------------------------------------
class CircularArray:
    def __init__(self, capacity):
        self.capacity = capacity
        self.size = 0
        self.dq = [0] * capacity
        self.f = 0
        self.r = capacity - 1

    def push_back(self, v):
        self.dq[self.r] = v
        self.r -= 1
        if self.r < 0:
            self.r = self.capacity - 1
        self.size += 1

    def pop_front(self):
        self.f -= 1
        if self.f < 0:
            self.f = self.capacity - 1
        self.size -= 1

    def pop_back(self):
        self.r = (self.r + 1) % self.capacity
        self.size -= 1

    def front(self):
        aux = self.f - 1
        if aux < 0:
            aux = self.capacity - 1
        return self.dq[aux]

    def back(self):
        aux = (self.r + 1) % self.capacity
        return self.dq[aux]

    def is_empty(self):
        return self.size == 0


def lower_upper_lemire(t, length, r, l, u):
    """DtwUtils.lowerUpperLemire (the array and the List version are the same code)."""
    du = CircularArray(2 * r + 2)
    dl = CircularArray(2 * r + 2)
    du.push_back(0)
    dl.push_back(0)
    for i in range(1, length):
        if i > r:
            u[i - r - 1] = t[du.front()]
            l[i - r - 1] = t[dl.front()]
        if t[i] > t[i - 1]:
            du.pop_back()
            while not du.is_empty() and t[i] > t[du.back()]:
                du.pop_back()
        else:
            dl.pop_back()
            while not dl.is_empty() and t[i] < t[dl.back()]:
                dl.pop_back()
        du.push_back(i)
        dl.push_back(i)
        if i == 2 * r + 1 + du.front():
            du.pop_front()
        elif i == 2 * r + 1 + dl.front():
            dl.pop_front()
    for i in range(length, length + r + 1):
        if i - r - 1 < 0:       # (Python would wrap a negative index; Java throws -- whenever length <= r)
            raise ArrayIndexOutOfBoundsException(str(i - r - 1))
        u[i - r - 1] = t[du.front()]
        l[i - r - 1] = t[dl.front()]
        if i - du.front() >= 2 * r + 1:
            du.pop_front()
        if i - dl.front() >= 2 * r + 1:
            dl.pop_front()


def _min(a, b):
    return a if a < b else b


def _div(a, b):
    """Java double division: x / 0.0 is +-Infinity or NaN, never an exception."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return float("nan")
        return math.copysign(float("inf"), a) * math.copysign(1.0, b)


def _dist(v1, v2):
    return (v1 - v2) * (v1 - v2)


def lb_kim_hierarchy(t, q, j, length, mean, std, bsf):
    x0 = _div(t[j] - mean, std)
    y0 = _div(t[length - 1 + j] - mean, std)
    lb = _dist(x0, q[0]) + _dist(y0, q[length - 1])
    if lb >= bsf:
        return lb
    x1 = _div(t[j + 1] - mean, std)
    d = _min(_dist(x1, q[0]), _dist(x0, q[1]))
    d = _min(d, _dist(x1, q[1]))
    lb += d
    if lb >= bsf:
        return lb
    y1 = _div(t[length - 2 + j] - mean, std)
    d = _min(_dist(y1, q[length - 1]), _dist(y0, q[length - 2]))
    d = _min(d, _dist(y1, q[length - 2]))
    lb += d
    if lb >= bsf:
        return lb
    x2 = _div(t[j + 2] - mean, std)
    d = _min(_dist(x0, q[2]), _dist(x1, q[2]))
    d = _min(d, _dist(x2, q[2]))
    d = _min(d, _dist(x2, q[1]))
    d = _min(d, _dist(x2, q[0]))
    lb += d
    if lb >= bsf:
        return lb
    y2 = _div(t[length - 3 + j] - mean, std)
    d = _min(_dist(y0, q[length - 3]), _dist(y1, q[length - 3]))
    d = _min(d, _dist(y2, q[length - 3]))
    d = _min(d, _dist(y2, q[length - 2]))
    d = _min(d, _dist(y2, q[length - 1]))
    lb += d
    return lb


def lb_keogh_cumulative(order, t, uo, lo, cb, j, length, mean, std, best_so_far):
    lb = 0.0
    i = 0
    while i < length and lb < best_so_far:
        x = _div(t[order[i] + j] - mean, std)
        d = 0.0
        if x > uo[i]:
            d = _dist(x, uo[i])
        elif x < lo[i]:
            d = _dist(x, lo[i])
        lb += d
        cb[order[i]] = d
        i += 1
    return lb


def lb_keogh_data_cumulative(order, qo, cb, I, l, u, length, mean, std, best_so_far):
    lb = 0.0
    i = 0
    while i < length and lb < best_so_far:
        uu = _div(u[order[i] + I] - mean, std)
        ll = _div(l[order[i] + I] - mean, std)
        d = 0.0
        if qo[i] > uu:
            d = _dist(qo[i], uu)
        else:
            if qo[i] < ll:
                d = _dist(qo[i], ll)
        lb += d
        cb[order[i]] = d
        i += 1
    return lb


def dtw(A, B, cb, m, r, bsf):
    cost = [INF] * (2 * r + 1)
    cost_prev = [INF] * (2 * r + 1)
    k = 0
    for i in range(m):
        k = max(0, r - i)
        min_cost = INF
        for j in range(max(0, i - r), min(m - 1, i + r) + 1):
            if i == 0 and j == 0:
                cost[k] = _dist(A[0], B[0])
                min_cost = cost[k]
                k += 1
                continue
            y = INF if (j - 1 < 0 or k - 1 < 0) else cost[k - 1]
            x = INF if (i - 1 < 0 or k + 1 > 2 * r) else cost_prev[k + 1]
            z = INF if (i - 1 < 0 or j - 1 < 0) else cost_prev[k]
            cost[k] = _min(_min(x, y), z) + _dist(A[i], B[j])
            if cost[k] < min_cost:
                min_cost = cost[k]
            k += 1
        if i + r < m - 1 and min_cost + cb[i + r + 1] >= bsf:
            return min_cost + cb[i + r + 1]
        cost, cost_prev = cost_prev, cost
    k -= 1
    return cost_prev[k]


# ---------------------------------------------------------------------------------------------------------------
def rsm_ed(series, query, left, right, last_segment, epsilon):
    n, length = len(series), len(query)
    answers = []
    cnt_candidate = 0
    for p in range(len(left)):
        cnt_candidate += right[p] - left[p] + 1
        begin = left[p] - (last_segment - 1) * WU0
        end = right[p] - (last_segment - 1) * WU0 + length - 1
        if begin < 1:
            begin = 1
        if end > n:
            end = n
        data = read_time_series(series, begin, end - begin + 1)
        i = 0
        while i + length - 1 < len(data):
            distance = 0.0
            j = 0
            while j < length and distance <= epsilon * epsilon:
                distance += (data[i + j] - query[j]) * (data[i + j] - query[j])
                j += 1
            if distance <= epsilon * epsilon:
                answers.append((begin + i, math.sqrt(distance)))
            i += 1
    return _sorted_answers(answers), cnt_candidate


def _query_stats(query):
    ex = ex2 = 0.0
    for value in query:
        ex += value
        ex2 += value * value
    mean_q = ex / len(query)
    std_q = math.sqrt(ex2 / len(query) - mean_q * mean_q)
    return mean_q, std_q


def _chunks(left, right, last_segment, query_length, n):
    """The greedy chunk merge shared by both cNSM engines: yields (beginIdx, endIdx, begin, end)."""
    # GUARD (not in the Java code; the oracle and the library have the same one): a list that is not sorted by left ends
    # in readTimeSeries' range check or in an IndexOutOfBoundsException in Java, depending on the list -- and Python's
    # negative indices would silently wrap.  The engines only ever pass sortAndMergeIntervals' output.
    if any(left[p] < left[p - 1] for p in range(1, len(left))):
        raise IllegalArgumentException("interval list not sorted by left")
    idx = 0
    size = len(left)
    while idx < size:
        begin_idx = end_idx = idx
        begin = left[idx] - (last_segment - 1) * WU0
        end = right[idx] - (last_segment - 1) * WU0 + query_length - 1
        if begin < 1:
            begin = 1
        length = end - begin + 1
        idx += 1
        while idx < size:
            begin = left[idx] - (last_segment - 1) * WU0
            new_length = length + begin - end - 1
            end = right[idx] - (last_segment - 1) * WU0 + query_length - 1
            if end > n:
                end = n
            new_length += end - begin + 1
            if new_length > MAX_SCAN_DATA_LENGTH:
                break
            end_idx = idx
            length = new_length
            idx += 1
        begin = left[begin_idx] - (last_segment - 1) * WU0
        end = right[end_idx] - (last_segment - 1) * WU0 + query_length - 1
        if begin < 1:
            begin = 1
        if end > n:
            end = n
        yield begin_idx, end_idx, begin, end


def cnsm_ed(series, query, left, right, last_segment, epsilon, alpha, beta):
    n, query_length = len(series), len(query)
    mean_q, std_q = _query_stats(query)
    answers = []
    cnt_candidate = 0
    zq = [_div(query[i] - mean_q, std_q) for i in range(query_length)]
    q_tmp = [(zq[i], i) for i in range(query_length)]
    # Arrays.sort(Q_tmp, (o1, o2) -> Double.compare(Math.abs(o2.value), Math.abs(o1.value))): stable, consistent
    q_tmp = sorted(q_tmp, key=functools.cmp_to_key(lambda o1, o2: _java_double_compare(abs(o2[0]), abs(o1[0]))))
    zq = [v for v, _ in q_tmp]
    order = [i for _, i in q_tmp]
    for begin_idx, end_idx, begin, end in _chunks(left, right, last_segment, query_length, n):
        data = read_time_series(series, begin, end - begin + 1)
        for idx1 in range(begin_idx, end_idx + 1):
            cnt_candidate += right[idx1] - left[idx1] + 1
            ex = ex2 = 0.0
            T = [0.0] * (2 * query_length)
            begin1 = left[idx1] - (last_segment - 1) * WU0 - begin
            end1 = right[idx1] - (last_segment - 1) * WU0 + query_length - 1 - begin
            if begin1 < 0:
                begin1 = 0
            if end1 > len(data) - 1:
                end1 = len(data) - 1
            for i in range(begin1, end1 + 1):
                d = data[i]
                ex += d
                ex2 += d * d
                T[i % query_length] = d
                T[(i % query_length) + query_length] = d
                if i - begin1 >= query_length - 1:
                    j = (i + 1) % query_length
                    mean = ex / query_length
                    v = ex2 / query_length - mean * mean
                    std = math.sqrt(v) if v >= 0 else float("nan")   # Math.sqrt of a negative is NaN
                    if abs(mean - mean_q) <= beta and _div(std, std_q) <= alpha and _div(std, std_q) >= _div(1.0, alpha):
                        dist = 0.0
                        k = 0
                        while k < query_length and dist <= epsilon * epsilon:
                            x = _div(T[order[k] + j] - mean, std)
                            dist += (x - zq[k]) * (x - zq[k])
                            k += 1
                        if dist <= epsilon * epsilon:
                            answers.append((begin + i - query_length + 1, math.sqrt(dist)))
                    ex -= T[j]
                    ex2 -= T[j] * T[j]
    return _sorted_answers(answers), cnt_candidate


def _dtw_decide(T, j, Q, order, qo, uo, lo, l_buff, u_buff, I, length, mean, std, rho, epsilon, normalise):
    """The cascade both DTW engines run per window once mean / std are known; returns the distance or None."""
    lb_kim = lb_kim_hierarchy(T, Q, j, length, mean, std, epsilon * epsilon)
    if lb_kim <= epsilon * epsilon:
        cb1 = [0.0] * length
        lb_k = lb_keogh_cumulative(order, T, uo, lo, cb1, j, length, mean, std, epsilon * epsilon)
        if lb_k <= epsilon * epsilon:
            if normalise:
                cT = [_div(T[k + j] - mean, std) for k in range(length)]   # NormQueryEngineDtw.java:564-567 (before lbK2)
            cb2 = [0.0] * length
            lb_k2 = lb_keogh_data_cumulative(order, qo, cb2, I, l_buff, u_buff, length, mean, std, epsilon * epsilon)
            if lb_k2 <= epsilon * epsilon:
                if not normalise:
                    cT = T[j: j + length]                               # System.arraycopy(T, j, cT, 0, length)
                cb = [0.0] * length
                src = cb1 if lb_k > lb_k2 else cb2
                cb[length - 1] = src[length - 1]
                for k in range(length - 2, -1, -1):
                    cb[k] = cb[k + 1] + src[k]
                dist = dtw(cT, Q, cb, length, rho, epsilon * epsilon)
                if dist <= epsilon * epsilon:
                    return dist
    return None


def rsm_dtw(series, query, left, right, last_segment, epsilon, rho):
    n, length = len(series), len(query)
    answers = []
    cnt_candidate = 0
    Q = [query[i] for i in range(length)]
    u, l = [0.0] * length, [0.0] * length
    lower_upper_lemire(Q, length, rho, l, u)
    order = list(range(length))           # the reorder is commented out (K/QueryEngineDtw.java:369-372): tmpQ stays as built
    qo = [Q[o] for o in order]
    uo = [u[o] for o in order]
    lo = [l[o] for o in order]
    for p in range(len(left)):
        cnt_candidate += right[p] - left[p] + 1
        begin = left[p] - (last_segment - 1) * WU0
        end = right[p] - (last_segment - 1) * WU0 + length - 1
        if begin < 1:
            begin = 1
        if end > n:
            end = n
        data = read_time_series(series, begin, end - begin + 1)
        T = [0.0] * (2 * length)
        u_buff, l_buff = [0.0] * len(data), [0.0] * len(data)
        lower_upper_lemire(data, len(data), rho, l_buff, u_buff)
        for i in range(len(data)):
            d = data[i]
            T[i % length] = d
            T[(i % length) + length] = d
            if i >= length - 1:
                j = (i + 1) % length
                dist = _dtw_decide(T, j, Q, order, qo, uo, lo, l_buff, u_buff, i - length + 1, length, 0, 1, rho, epsilon, False)
                if dist is not None:
                    answers.append((begin + i - length + 1, math.sqrt(dist)))
    return _sorted_answers(answers), cnt_candidate


def _java_int(v):
    """(int) of a double: toward zero, NaN -> 0, saturating."""
    if v != v:
        return 0
    if v >= 2147483647.0:
        return 2147483647
    if v <= -2147483648.0:
        return -2147483648
    return int(v)


def cnsm_dtw(series, query, left, right, last_segment, epsilon, rho, alpha, beta):
    n, query_length = len(series), len(query)
    mean_q, std_q = _query_stats(query)
    answers = []
    cnt_candidate = 0
    zq = [_div(query[i] - mean_q, std_q) for i in range(query_length)]
    u, l = [0.0] * query_length, [0.0] * query_length
    lower_upper_lemire(zq, query_length, rho, l, u)
    tmp_q = [(zq[i], i) for i in range(query_length)]
    try:
        java_arrays_sort(tmp_q, lambda o1, o2: _java_int(abs(o2[0]) - abs(o1[0])))
    except ComparisonContractViolation as e:
        raise IllegalArgumentException(str(e))
    order = [i for _, i in tmp_q]
    qo = [zq[o] for o in order]
    uo = [u[o] for o in order]
    lo = [l[o] for o in order]
    for begin_idx, end_idx, begin, end in _chunks(left, right, last_segment, query_length, n):
        data = read_time_series(series, begin, end - begin + 1)
        u_buff, l_buff = [0.0] * len(data), [0.0] * len(data)
        lower_upper_lemire(data, len(data), rho, l_buff, u_buff)
        for idx1 in range(begin_idx, end_idx + 1):
            cnt_candidate += right[idx1] - left[idx1] + 1
            ex = ex2 = 0.0
            T = [0.0] * (2 * query_length)
            begin1 = left[idx1] - (last_segment - 1) * WU0 - begin
            end1 = right[idx1] - (last_segment - 1) * WU0 + query_length - 1 - begin
            if begin1 < 0:
                begin1 = 0
            if end1 > len(data) - 1:
                end1 = len(data) - 1
            for i in range(begin1, end1 + 1):
                d = data[i]
                ex += d
                ex2 += d * d
                T[i % query_length] = d
                T[(i % query_length) + query_length] = d
                if i - begin1 >= query_length - 1:
                    j = (i + 1) % query_length
                    mean = ex / query_length
                    v = ex2 / query_length - mean * mean
                    std = math.sqrt(v) if v >= 0 else float("nan")
                    if abs(mean - mean_q) <= beta and _div(std, std_q) <= alpha and _div(std, std_q) >= _div(1.0, alpha):
                        dist = _dtw_decide(T, j, zq, order, qo, uo, lo, l_buff, u_buff, i - query_length + 1, query_length,
                                           mean, std, rho, epsilon, True)
                        if dist is not None:
                            answers.append((begin + i - query_length + 1, math.sqrt(dist)))
                    ex -= T[j]
                    ex2 -= T[j] * T[j]
    return _sorted_answers(answers), cnt_candidate


# ---------------------------------------------------------------------------------------------------------------
# The index-free full-scan executors of the reference's experiments (U/ = K/experiments/ucr/), written from the Java text
# like the engines above.  The scanner (an iterator over TimeSeriesNodes) is restated as a pass over the first N points:
# nextData() (U/UcrEdQueryExecutor.java:57-79) counts `++cnt <= N` only inside a node, so N is a loose upper limit; the
# callers pass N = the length of the data.
UCR_EPOCH = 100000              # U/UcrDtwQueryExecutor.java:97


def ucr_ed(series, query, epsilon, alpha, beta):
    """U/UcrEdQueryExecutor.java:96-175.  One pair of running sums over the whole scan; 1-based answer offsets
    (`i - M + 2`, :161).  distance() (:85-92) loops `while sum < bsf`: a partial sum that lands exactly on bsf ends it."""
    M = len(query)
    ex = ex2 = 0.0
    Q = [0.0] * M
    for i in range(M):
        d = query[i]
        ex += d
        ex2 += d * d
        Q[i] = d
    mean_q = ex / M
    std_q = ex2 / M
    v = std_q - mean_q * mean_q
    std_q = math.sqrt(v) if v >= 0 else float("nan")
    for i in range(M):
        Q[i] = _div(Q[i] - mean_q, std_q)
    q_tmp = [(Q[i], i) for i in range(M)]
    # Arrays.sort with Double.compare(|o2|, |o1|): a consistent comparator, so the stable sort's result is determined by it
    q_tmp = sorted(q_tmp, key=functools.cmp_to_key(lambda o1, o2: _java_double_compare(abs(o2[0]), abs(o1[0]))))
    order = [0] * M
    for i in range(M):
        Q[i] = q_tmp[i][0]
        order[i] = q_tmp[i][1]
    t = [0.0] * (2 * M)
    bsf = epsilon * epsilon
    ex = ex2 = 0.0
    answers = []
    i = 0
    for d in series:
        ex += d
        ex2 += d * d
        t[i % M] = d
        t[(i % M) + M] = d
        if i >= M - 1:
            j = (i + 1) % M
            mean = ex / M
            std = ex2 / M
            v = std - mean * mean
            std = math.sqrt(v) if v >= 0 else float("nan")
            if abs(mean - mean_q) <= beta and _div(std, std_q) <= alpha and _div(std, std_q) >= _div(1.0, alpha):
                total = 0.0                                  # distance(), :85-92
                k = 0
                while k < M and total < bsf:
                    x = _div(t[order[k] + j] - mean, std)
                    total += (x - Q[k]) * (x - Q[k])
                    k += 1
                if total <= epsilon * epsilon:
                    answers.append((i - M + 2, math.sqrt(total)))
            ex -= t[j]
            ex2 -= t[j] * t[j]
        i += 1
    return _sorted_answers(answers)


def ucr_dtw(series, query, epsilon, rho, alpha, beta):
    """U/UcrDtwQueryExecutor.java:84-315.  The data in buffers of EPOCH points that overlap by M - 1 (:175-203); running
    sums restart per buffer (:215-216); the data envelope is built over the buffer (:210); cb1 / cb2 are allocated once and
    never cleared between windows (:157-161 -- the engines allocate fresh ones); 0-BASED answer offsets (:283); the query
    is sorted with the contract-breaking comparator (int)(|o2| - |o1|) (:139-142) like NormQueryEngineDtw."""
    M = len(query)
    ex = ex2 = 0.0
    q = [0.0] * M
    for i in range(M):
        d = query[i]
        ex += d
        ex2 += d * d
        q[i] = d
    mean_q = ex / M
    std_q = ex2 / M
    v = std_q - mean_q * mean_q
    std_q = math.sqrt(v) if v >= 0 else float("nan")
    for i in range(M):
        q[i] = _div(q[i] - mean_q, std_q)
    l, u = [0.0] * M, [0.0] * M
    lower_upper_lemire(q, M, rho, l, u)
    q_tmp = [(q[i], i) for i in range(M)]
    try:
        java_arrays_sort(q_tmp, lambda o1, o2: _java_int(abs(o2[0]) - abs(o1[0])))
    except ComparisonContractViolation as e:
        raise IllegalArgumentException(str(e))
    order = [o for _, o in q_tmp]
    qo = [q[o] for o in order]
    uo = [u[o] for o in order]
    lo = [l[o] for o in order]
    cb, cb1, cb2 = [0.0] * M, [0.0] * M, [0.0] * M
    t = [0.0] * (2 * M)
    tz = [0.0] * M
    buffer = [0.0] * UCR_EPOCH
    u_buff, l_buff = [0.0] * UCR_EPOCH, [0.0] * UCR_EPOCH
    source = iter(series)
    bsf = epsilon * epsilon
    answers = []
    done = False
    it = 0
    while not done:
        if it == 0:
            for k in range(M - 1):
                d = next(source, None)
                if d is not None:                          # (EOFException: do nothing)
                    buffer[k] = d
        else:
            for k in range(M - 1):
                buffer[k] = buffer[UCR_EPOCH - M + 1 + k]
        ep = M - 1
        while ep < UCR_EPOCH:
            d = next(source, None)
            if d is None:
                break
            buffer[ep] = d
            ep += 1
        if ep <= M - 1:
            done = True
        else:
            lower_upper_lemire(buffer, ep, rho, l_buff, u_buff)
            ex = ex2 = 0.0
            for i in range(ep):
                d = buffer[i]
                ex += d
                ex2 += d * d
                t[i % M] = d
                t[(i % M) + M] = d
                if i >= M - 1:
                    mean = ex / M
                    std = ex2 / M
                    v = std - mean * mean
                    std = math.sqrt(v) if v >= 0 else float("nan")
                    j = (i + 1) % M
                    I = i - (M - 1)
                    if abs(mean - mean_q) <= beta and _div(std, std_q) <= alpha and _div(std, std_q) >= _div(1.0, alpha):
                        lb_kim = lb_kim_hierarchy(t, q, j, M, mean, std, bsf)
                        if lb_kim <= bsf:
                            lb_k = lb_keogh_cumulative(order, t, uo, lo, cb1, j, M, mean, std, bsf)
                            if lb_k <= bsf:
                                for k in range(M):
                                    tz[k] = _div(t[k + j] - mean, std)
                                lb_k2 = lb_keogh_data_cumulative(order, qo, cb2, I, l_buff, u_buff, M, mean, std, bsf)
                                if lb_k2 <= bsf:
                                    src = cb1 if lb_k > lb_k2 else cb2
                                    cb[M - 1] = src[M - 1]
                                    for k in range(M - 2, -1, -1):
                                        cb[k] = cb[k + 1] + src[k]
                                    dist = dtw(tz, q, cb, M, rho, bsf)
                                    if dist <= bsf:
                                        answers.append((it * (UCR_EPOCH - M + 1) + i - M + 1, math.sqrt(dist)))
                    ex -= t[j]
                    ex2 -= t[j] * t[j]
            if ep < UCR_EPOCH:
                done = True
            else:
                it += 1
    return _sorted_answers(answers)


# ---------------------------------------------------------------------------------------------------------------
# java.util.TimSort (OpenJDK 8, the JDK the reference's pom.xml targets: <source>1.8</source>), restated from its
# published source: Arrays.sort(T[], Comparator) -> TimSort.sort(a, 0, a.length, c, null, 0, 0).  The JDK is a dependency
# that is absent from /root/reference; it matters here because NormQueryEngineDtw sorts with a comparator that violates
# the Comparator contract ((int)(|o2| - |o1|), K/NormQueryEngineDtw.java:475-478): the resulting permutation is then a
# property of this algorithm -- its run detection, binary insertion, merge order and galloping -- not of the comparator.
# In particular a query whose |zQ| never rises by >= 1 between neighbours is ONE non-descending run and comes back
# unchanged, and the sort can end in IllegalArgumentException("Comparison method violates its general contract!").
class ComparisonContractViolation(Exception):  # noqa: E302
    """java.lang.IllegalArgumentException("Comparison method violates its general contract!") out of mergeLo / mergeHi."""


_MIN_MERGE = 32
_MIN_GALLOP = 7


def _count_run_and_make_ascending(a, lo, hi, c):
    run_hi = lo + 1
    if run_hi == hi:
        return 1
    run_hi += 1
    if c(a[run_hi - 1], a[lo]) < 0:                       # descending
        while run_hi < hi and c(a[run_hi], a[run_hi - 1]) < 0:
            run_hi += 1
        i, j = lo, run_hi - 1                             # reverseRange(a, lo, runHi)
        while i < j:
            a[i], a[j] = a[j], a[i]
            i += 1
            j -= 1
    else:                                                 # ascending
        while run_hi < hi and c(a[run_hi], a[run_hi - 1]) >= 0:
            run_hi += 1
    return run_hi - lo


def _binary_sort(a, lo, hi, start, c):
    if start == lo:
        start += 1
    while start < hi:
        pivot = a[start]
        left, right = lo, start
        while left < right:
            mid = (left + right) >> 1
            if c(pivot, a[mid]) < 0:
                right = mid
            else:
                left = mid + 1
        n = start - left
        a[left + 1: left + 1 + n] = a[left: left + n]     # System.arraycopy(a, left, a, left + 1, n)
        a[left] = pivot
        start += 1


def _min_run_length(n):
    r = 0
    while n >= _MIN_MERGE:
        r |= (n & 1)
        n >>= 1
    return n + r


def _gallop_left(key, a, base, length, hint, c):
    last_ofs, ofs = 0, 1
    if c(key, a[base + hint]) > 0:
        max_ofs = length - hint
        while ofs < max_ofs and c(key, a[base + hint + ofs]) > 0:
            last_ofs = ofs
            ofs = (ofs << 1) + 1
        if ofs > max_ofs:
            ofs = max_ofs
        last_ofs += hint
        ofs += hint
    else:
        max_ofs = hint + 1
        while ofs < max_ofs and c(key, a[base + hint - ofs]) <= 0:
            last_ofs = ofs
            ofs = (ofs << 1) + 1
        if ofs > max_ofs:
            ofs = max_ofs
        last_ofs, ofs = hint - ofs, hint - last_ofs
    last_ofs += 1
    while last_ofs < ofs:
        m = last_ofs + ((ofs - last_ofs) >> 1)
        if c(key, a[base + m]) > 0:
            last_ofs = m + 1
        else:
            ofs = m
    return ofs


def _gallop_right(key, a, base, length, hint, c):
    ofs, last_ofs = 1, 0
    if c(key, a[base + hint]) < 0:
        max_ofs = hint + 1
        while ofs < max_ofs and c(key, a[base + hint - ofs]) < 0:
            last_ofs = ofs
            ofs = (ofs << 1) + 1
        if ofs > max_ofs:
            ofs = max_ofs
        last_ofs, ofs = hint - ofs, hint - last_ofs
    else:
        max_ofs = length - hint
        while ofs < max_ofs and c(key, a[base + hint + ofs]) >= 0:
            last_ofs = ofs
            ofs = (ofs << 1) + 1
        if ofs > max_ofs:
            ofs = max_ofs
        last_ofs += hint
        ofs += hint
    last_ofs += 1
    while last_ofs < ofs:
        m = last_ofs + ((ofs - last_ofs) >> 1)
        if c(key, a[base + m]) < 0:
            ofs = m
        else:
            last_ofs = m + 1
    return ofs


class _TimSort:
    def __init__(self, a, c):
        self.a, self.c = a, c
        self.min_gallop = _MIN_GALLOP
        self.run_base, self.run_len = [], []

    def push_run(self, base, length):
        self.run_base.append(base)
        self.run_len.append(length)

    def merge_collapse(self):                              # the JDK 8 form
        rl = self.run_len
        while len(rl) > 1:
            n = len(rl) - 2
            if n > 0 and rl[n - 1] <= rl[n] + rl[n + 1]:
                if rl[n - 1] < rl[n + 1]:
                    n -= 1
                self.merge_at(n)
            elif rl[n] <= rl[n + 1]:
                self.merge_at(n)
            else:
                break

    def merge_force_collapse(self):
        rl = self.run_len
        while len(rl) > 1:
            n = len(rl) - 2
            if n > 0 and rl[n - 1] < rl[n + 1]:
                n -= 1
            self.merge_at(n)

    def merge_at(self, i):
        a, c = self.a, self.c
        base1, len1 = self.run_base[i], self.run_len[i]
        base2, len2 = self.run_base[i + 1], self.run_len[i + 1]
        self.run_len[i] = len1 + len2
        del self.run_base[i + 1]                           # (slides run i+2 over when i is the 3rd-last run; pops otherwise)
        del self.run_len[i + 1]
        k = _gallop_right(a[base2], a, base1, len1, 0, c)
        base1 += k
        len1 -= k
        if len1 == 0:
            return
        len2 = _gallop_left(a[base1 + len1 - 1], a, base2, len2, len2 - 1, c)
        if len2 == 0:
            return
        if len1 <= len2:
            self.merge_lo(base1, len1, base2, len2)
        else:
            self.merge_hi(base1, len1, base2, len2)

    def merge_lo(self, base1, len1, base2, len2):
        a, c = self.a, self.c
        tmp = a[base1: base1 + len1]
        cursor1, cursor2, dest = 0, base2, base1
        a[dest] = a[cursor2]
        dest += 1
        cursor2 += 1
        len2 -= 1
        if len2 == 0:
            a[dest: dest + len1] = tmp[cursor1: cursor1 + len1]
            return
        if len1 == 1:
            a[dest: dest + len2] = a[cursor2: cursor2 + len2]
            a[dest + len2] = tmp[cursor1]
            return
        min_gallop = self.min_gallop
        done = False
        while not done:
            count1 = count2 = 0
            while True:
                if c(a[cursor2], tmp[cursor1]) < 0:
                    a[dest] = a[cursor2]
                    dest += 1
                    cursor2 += 1
                    count2 += 1
                    count1 = 0
                    len2 -= 1
                    if len2 == 0:
                        done = True
                        break
                else:
                    a[dest] = tmp[cursor1]
                    dest += 1
                    cursor1 += 1
                    count1 += 1
                    count2 = 0
                    len1 -= 1
                    if len1 == 1:
                        done = True
                        break
                if not ((count1 | count2) < min_gallop):
                    break
            if done:
                break
            while True:
                count1 = _gallop_right(a[cursor2], tmp, cursor1, len1, 0, c)
                if count1 != 0:
                    a[dest: dest + count1] = tmp[cursor1: cursor1 + count1]
                    dest += count1
                    cursor1 += count1
                    len1 -= count1
                    if len1 <= 1:
                        done = True
                        break
                a[dest] = a[cursor2]
                dest += 1
                cursor2 += 1
                len2 -= 1
                if len2 == 0:
                    done = True
                    break
                count2 = _gallop_left(tmp[cursor1], a, cursor2, len2, 0, c)
                if count2 != 0:
                    a[dest: dest + count2] = a[cursor2: cursor2 + count2]
                    dest += count2
                    cursor2 += count2
                    len2 -= count2
                    if len2 == 0:
                        done = True
                        break
                a[dest] = tmp[cursor1]
                dest += 1
                cursor1 += 1
                len1 -= 1
                if len1 == 1:
                    done = True
                    break
                min_gallop -= 1
                if not (count1 >= _MIN_GALLOP or count2 >= _MIN_GALLOP):
                    break
            if done:
                break
            if min_gallop < 0:
                min_gallop = 0
            min_gallop += 2
        self.min_gallop = 1 if min_gallop < 1 else min_gallop
        if len1 == 1:
            a[dest: dest + len2] = a[cursor2: cursor2 + len2]
            a[dest + len2] = tmp[cursor1]
        elif len1 == 0:
            raise ComparisonContractViolation("Comparison method violates its general contract!")
        else:
            a[dest: dest + len1] = tmp[cursor1: cursor1 + len1]

    def merge_hi(self, base1, len1, base2, len2):
        a, c = self.a, self.c
        tmp = a[base2: base2 + len2]
        tmp_base = 0
        cursor1 = base1 + len1 - 1
        cursor2 = tmp_base + len2 - 1
        dest = base2 + len2 - 1
        a[dest] = a[cursor1]
        dest -= 1
        cursor1 -= 1
        len1 -= 1
        if len1 == 0:
            a[dest - (len2 - 1): dest + 1] = tmp[tmp_base: tmp_base + len2]
            return
        if len2 == 1:
            dest -= len1
            cursor1 -= len1
            a[dest + 1: dest + 1 + len1] = a[cursor1 + 1: cursor1 + 1 + len1]
            a[dest] = tmp[cursor2]
            return
        min_gallop = self.min_gallop
        done = False
        while not done:
            count1 = count2 = 0
            while True:
                if c(tmp[cursor2], a[cursor1]) < 0:
                    a[dest] = a[cursor1]
                    dest -= 1
                    cursor1 -= 1
                    count1 += 1
                    count2 = 0
                    len1 -= 1
                    if len1 == 0:
                        done = True
                        break
                else:
                    a[dest] = tmp[cursor2]
                    dest -= 1
                    cursor2 -= 1
                    count2 += 1
                    count1 = 0
                    len2 -= 1
                    if len2 == 1:
                        done = True
                        break
                if not ((count1 | count2) < min_gallop):
                    break
            if done:
                break
            while True:
                count1 = len1 - _gallop_right(tmp[cursor2], a, base1, len1, len1 - 1, c)
                if count1 != 0:
                    dest -= count1
                    cursor1 -= count1
                    len1 -= count1
                    a[dest + 1: dest + 1 + count1] = a[cursor1 + 1: cursor1 + 1 + count1]
                    if len1 == 0:
                        done = True
                        break
                a[dest] = tmp[cursor2]
                dest -= 1
                cursor2 -= 1
                len2 -= 1
                if len2 == 1:
                    done = True
                    break
                count2 = len2 - _gallop_left(a[cursor1], tmp, tmp_base, len2, len2 - 1, c)
                if count2 != 0:
                    dest -= count2
                    cursor2 -= count2
                    len2 -= count2
                    a[dest + 1: dest + 1 + count2] = tmp[cursor2 + 1: cursor2 + 1 + count2]
                    if len2 <= 1:
                        done = True
                        break
                a[dest] = a[cursor1]
                dest -= 1
                cursor1 -= 1
                len1 -= 1
                if len1 == 0:
                    done = True
                    break
                min_gallop -= 1
                if not (count1 >= _MIN_GALLOP or count2 >= _MIN_GALLOP):
                    break
            if done:
                break
            if min_gallop < 0:
                min_gallop = 0
            min_gallop += 2
        self.min_gallop = 1 if min_gallop < 1 else min_gallop
        if len2 == 1:
            dest -= len1
            cursor1 -= len1
            a[dest + 1: dest + 1 + len1] = a[cursor1 + 1: cursor1 + 1 + len1]
            a[dest] = tmp[cursor2]
        elif len2 == 0:
            raise ComparisonContractViolation("Comparison method violates its general contract!")
        else:
            a[dest - (len2 - 1): dest + 1] = tmp[tmp_base: tmp_base + len2]


def java_arrays_sort(a, c):
    """Arrays.sort(T[] a, Comparator c) of JDK 8, in place on the list `a`; may raise ComparisonContractViolation."""
    lo, hi = 0, len(a)
    n_remaining = hi - lo
    if n_remaining < 2:
        return a
    if n_remaining < _MIN_MERGE:
        init_run_len = _count_run_and_make_ascending(a, lo, hi, c)
        _binary_sort(a, lo, hi, lo + init_run_len, c)
        return a
    ts = _TimSort(a, c)
    min_run = _min_run_length(n_remaining)
    while True:
        run_len = _count_run_and_make_ascending(a, lo, hi, c)
        if run_len < min_run:
            force = n_remaining if n_remaining <= min_run else min_run
            _binary_sort(a, lo, lo + force, lo + run_len, c)
            run_len = force
        ts.push_run(lo, run_len)
        ts.merge_collapse()
        lo += run_len
        n_remaining -= run_len
        if n_remaining == 0:
            break
    ts.merge_force_collapse()
    return a


# ---------------------------------------------------------------------------------------------------------------
# Index build: IndexBuilder.SingleIndexBuilder.run (K/IndexBuilder.java:186-347), IndexNodeUtils.mergeIndexNode /
# addInterval (K/utils/IndexNodeUtils.java:30-90), IndexNode.toBytesCompact / getStatisticInfoPair
# (K/common/entity/IndexNode.java:61-100, :145-158), MeanIntervalUtils.toRound (K/utils/MeanIntervalUtils.java:50-60),
# ByteUtils.listTripleToByteArray / listIntToByteArray (K/utils/ByteUtils.java), IndexFileOperator.writeAll
# (K/operator/file/IndexFileOperator.java:127-165).  Returns the bytes of files/index-{N}-{Wu}.
import struct

MAXIMUM_DIFF = 256  # IndexNode.MAXIMUM_DIFF


def _double_key(v):
    """what HashMap<Double, .> / Double.equals distinguish: doubleToLongBits (one NaN, 0.0 != -0.0)"""
    return struct.pack(">d", float("nan") if v != v else v)


def _compare_to(a, b):
    return _java_double_compare(a, b)


def to_round(value):
    value *= math.pow(10, 2 - 1)
    # Math.floor returns a double: floor(-0.0) is -0.0 (Python's math.floor returns the int 0 and loses the sign), NaN and
    # the infinities come back unchanged
    int_value = float(math.floor(value)) if value == value and abs(value) != float("inf") and value != 0.0 else value
    diff = value - int_value
    ret = float(int_value)
    if _java_double_compare(diff, 0.5) >= 0:
        ret += 0.5
    ret *= math.pow(10, -2 + 1)
    return ret


def _add_interval(node, position):
    while position[1] - position[0] >= MAXIMUM_DIFF:
        new_first = position[0] + MAXIMUM_DIFF - 1
        node.append([position[0], new_first])
        position[0] = new_first + 1
    node.append(position)


def merge_index_node(node1, node2):
    ret = []
    index1 = index2 = 0
    last1 = last2 = None
    while index1 < len(node1) and index2 < len(node2):
        position1, position2 = node1[index1], node2[index2]
        if last1 is None:
            last1 = [position1[0], position1[1]]
        if last2 is None:
            last2 = [position2[0], position2[1]]
        if last1[1] + 1 < last2[0]:
            _add_interval(ret, last1)
            index1 += 1
            last1 = None
        elif last2[1] + 1 < last1[0]:
            _add_interval(ret, last2)
            index2 += 1
            last2 = None
        else:
            if last1[1] < last2[1]:
                if last1[0] < last2[0]:
                    last2[0] = last1[0]
                index1 += 1
                last1 = None
            else:
                if last2[0] < last1[0]:
                    last1[0] = last2[0]
                index2 += 1
                last2 = None
    for i in range(index1, len(node1)):
        if last1 is None:
            last1 = [node1[i][0], node1[i][1]]
        _add_interval(ret, last1)
        last1 = None
    for i in range(index2, len(node2)):
        if last2 is None:
            last2 = [node2[i][0], node2[i][1]]
        _add_interval(ret, last2)
        last2 = None
    return ret


def _java_byte(v):
    """(byte) of an int"""
    return v & 0xFF


def to_bytes_compact(positions):
    result = bytearray(4 * len(positions) * 2)
    index = length = count = 0
    is_packing = False
    while index < len(positions):
        if not is_packing:
            result[length: length + 4] = struct.pack(">i", positions[index][0])
            length += 4 + 1
            diff = positions[index][1] - positions[index][0]
            result[length] = _java_byte(diff - 128)
            length += 1
            is_packing = True
            count = 1
        else:
            diff = positions[index][0] - positions[index - 1][1]
            if diff < MAXIMUM_DIFF and (count - 1) // 2 + 2 < MAXIMUM_DIFF:
                result[length] = _java_byte(diff - 128)
                length += 1
                diff = positions[index][1] - positions[index][0]
                result[length] = _java_byte(diff - 128)
                length += 1
                count += 2
            else:
                result[length - count - 1] = _java_byte((count - 1) // 2 - 128)
                is_packing = False
                continue
        index += 1
    if is_packing:
        result[length - count - 1] = _java_byte((count - 1) // 2 - 128)
    return bytes(result[:length])


def parse_bytes_compact(data):
    """IndexNode.parseBytesCompact (K/common/entity/IndexNode.java:115-133)"""
    positions = []
    index = 0

    def sbyte(b):
        return b - 256 if b >= 128 else b
    while index < len(data):
        left = struct.unpack(">i", data[index: index + 4])[0]
        index += 4
        count = sbyte(data[index]) + 128
        index += 1
        right = left + sbyte(data[index]) + 128
        index += 1
        positions.append((left, right))
        for _ in range(count):
            left = right + sbyte(data[index]) + 128
            index += 1
            right = left + sbyte(data[index]) + 128
            index += 1
            positions.append((left, right))
    return positions


def index_build_file(series, w, epoch=100000, canonical_nan_key=True):
    """IndexBuilder.run + the index file image.  `canonical_nan_key`: Java keeps ONE NaN row (Double.equals) and writes the raw
    bits (HBase Bytes.toBytes(double) = doubleToRawLongBits) of the first NaN key it met -- the hardware's default NaN when
    it comes out of inf - inf (sign bit set on x86, clear on AArch64): not a function of the Java text.  The oracle and the
    library write the canonical 0x7ff8000000000000; True does the same here, False keeps what this
    machine's arithmetic produced."""
    n = len(series)
    it = 0
    done = False
    nxt = 0                         # the scanner: nextData() / getCurrentData()
    buffer = [0.0] * epoch
    t = [0.0] * (2 * w)
    last_mean_round = None
    index_node = None
    index_node_map = {}             # HashMap<Double, IndexNode>: key bits -> (key, positions)
    while not done:
        if it == 0:
            for k in range(w - 1):
                if nxt < n:
                    buffer[k] = series[nxt]
                    nxt += 1
        else:
            for k in range(w - 1):
                buffer[k] = buffer[epoch - w + 1 + k]
        ep = w - 1
        while ep < epoch:
            if nxt < n:
                buffer[ep] = series[nxt]
                nxt += 1
                ep += 1
            else:
                break
        if ep <= w - 1:
            done = True
        else:
            ex = ex2 = 0.0
            for i in range(ep):
                d = buffer[i]
                ex += d
                ex2 += d * d
                t[i % w] = d
                t[(i % w) + w] = d
                if i >= w - 1:
                    mean = _div(ex, w)
                    j = (i + 1) % w
                    loc = it * (epoch - w + 1) + i - w + 1 + 1
                    if loc > n:
                        done = True
                        break
                    cur_mean_round = to_round(mean)
                    if canonical_nan_key and cur_mean_round != cur_mean_round:
                        cur_mean_round = float("nan")       # 0x7ff8000000000000
                    if last_mean_round is None or _double_key(last_mean_round) != _double_key(cur_mean_round) \
                            or loc - index_node[-1][0] == MAXIMUM_DIFF - 1:
                        if last_mean_round is not None:      # HashMap.put: an equal key that is already there stays
                            kk = _double_key(last_mean_round)
                            index_node_map[kk] = (index_node_map[kk][0] if kk in index_node_map else last_mean_round, index_node)
                        got = index_node_map.get(_double_key(cur_mean_round))
                        index_node = got[1] if got is not None else []
                        index_node.append([loc, loc])
                        last_mean_round = cur_mean_round
                    else:
                        index_node[-1][1] = loc
                    ex -= t[j]
                    ex2 -= t[j] * t[j]
            if ep < epoch:
                done = True
            else:
                it += 1
    if index_node is not None and len(index_node) > 0:
        kk = _double_key(last_mean_round)
        index_node_map[kk] = (index_node_map[kk][0] if kk in index_node_map else last_mean_round, index_node)
    # step 2
    raw = []
    total = 0.0
    for key, node in index_node_map.values():
        raw.append((key, len(node)))
        total += len(node)
    average = total / len(raw)                        # StatisticInfo.getAverage (ints sum exactly)
    raw.sort(key=functools.cmp_to_key(lambda o1, o2: -_compare_to(o1[0], o2[0])))
    index_store = []                                  # TreeMap<Double, IndexNode>, filled below, sorted at the end
    statistic_info = []
    last = index_node_map[_double_key(raw[0][0])][1]
    for i in range(1, len(raw)):
        current = index_node_map[_double_key(raw[i][0])][1]
        is_merged = False
        if raw[i][1] < average * 1.2:
            merged = merge_index_node(last, current)
            if len(merged) < (len(last) + len(current)) * 0.8:
                last = merged
                is_merged = True
        if not is_merged:
            key = raw[i - 1][0]
            index_store.append((key, last))
            statistic_info.append([key, len(last), sum(p[1] - p[0] + 1 for p in last)])
            last = current
    key = raw[-1][0]
    index_store.append((key, last))
    statistic_info.append([key, len(last), sum(p[1] - p[0] + 1 for p in last)])
    index_store.sort(key=functools.cmp_to_key(lambda a, b: _compare_to(a[0], b[0])))
    # IndexFileOperator.writeAll
    out = bytearray()
    offsets = []
    for key, node in index_store:
        offsets.append(len(out))
        out += struct.pack(">d", key) + to_bytes_compact(node)
    offsets.append(len(out))
    statistic_info.sort(key=functools.cmp_to_key(lambda a, b: _java_double_compare(a[0], b[0])))
    for i in range(1, len(statistic_info)):
        statistic_info[i][1] += statistic_info[i - 1][1]
        statistic_info[i][2] += statistic_info[i - 1][2]
    for key, a, b in statistic_info:
        out += struct.pack(">dii", key, a, b)
    offsets.append(len(out))
    for o in offsets:
        out += struct.pack(">i", o)
    return bytes(out), dict(rows_before=len(raw), rows_after=len(index_store))


def read_index_file(data):
    """IndexFileOperator.readOffsetInfo / readIndexes(all) / readStatisticInfo on an in-memory image:
    (rows = [(key, positions)], statistic triples, offsets)."""
    last_line_offset = struct.unpack(">i", data[-4:])[0]
    offsets = list(struct.unpack(">%di" % ((len(data) - last_line_offset) // 4), data[last_line_offset:]))
    rows = []
    for i in range(len(offsets) - 2):
        line = data[offsets[i]: offsets[i + 1]]
        rows.append((struct.unpack(">d", line[:8])[0], parse_bytes_compact(line[8:])))
    st = data[offsets[-2]: offsets[-1]]
    triples = [struct.unpack(">dii", st[k: k + 16]) for k in range(0, len(st), 16)]
    return rows, triples, offsets


def sort_and_merge_intervals(intervals):
    """QueryEngine.sortAndMergeIntervals (K/QueryEngine.java:664-693) on a list of (left, right, epsilon)."""
    if len(intervals) <= 1:
        return list(intervals)
    intervals = sorted(intervals, key=lambda iv: iv[0])           # List.sort(comparingInt(getLeft)): stable
    start, end, epsilon = intervals[0]
    result = []
    for current in intervals[1:]:
        if current[0] - 1 <= end:
            end = max(current[1], end)
            epsilon = min(current[2], epsilon)
        else:
            result.append((start, end, epsilon))
            start, end, epsilon = current
    result.append((start, end, epsilon))
    return result
