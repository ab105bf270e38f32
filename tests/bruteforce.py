"""Independent brute-force checkers (numpy / plain Python), written from the problem definitions rather than
from the reference's loops, used to cross-check the C oracle (oracle/kvm_oracle.c).

  * ED / z-normalised ED by direct formula (cf. the semantics of K/CsvTester.java:58-114),
  * banded DTW on the full m x m matrix with the Sakoe-Chiba constraint |i-j| <= r,
  * clipped sliding max/min (what DtwUtils.lowerUpperLemire computes).
"""
import numpy as np


def window_starts(left, right, shift, n, m):
    out = []
    for l, r in zip(left, right):
        lo = max(1, int(l) - shift)
        hi = min(int(r) - shift, n - m + 1)
        out.extend(range(lo, hi + 1))
    return np.asarray(out, dtype=np.int64)


def ed2(series, q, start):
    w = series[start - 1: start - 1 + len(q)]
    d = w - q
    return float(np.dot(d, d))


def znorm_stats(w):
    m = len(w)
    mean = w.sum() / m
    std = np.sqrt((w * w).sum() / m - mean * mean)
    return mean, std


def banded_dtw_full(A, B, r):
    """O(m^2) memory DP; returns cost[m-1][m-1] with the band |i-j| <= r."""
    m = len(A)
    INF = float("inf")
    D = np.full((m, m), INF)
    for i in range(m):
        lo, hi = max(0, i - r), min(m - 1, i + r)
        for j in range(lo, hi + 1):
            c = (A[i] - B[j]) * (A[i] - B[j])
            if i == 0 and j == 0:
                D[i, j] = c
                continue
            best = INF
            if i > 0 and abs(i - 1 - j) <= r:
                best = min(best, D[i - 1, j])
            if j > 0 and abs(i - (j - 1)) <= r:
                best = min(best, D[i, j - 1])
            if i > 0 and j > 0:
                best = min(best, D[i - 1, j - 1])
            D[i, j] = best + c
    return float(D[m - 1, m - 1])


def envelope_naive(t, r):
    n = len(t)
    u = np.empty(n)
    l = np.empty(n)
    for i in range(n):
        lo, hi = max(0, i - r), min(n - 1, i + r)
        u[i] = t[lo:hi + 1].max()
        l[i] = t[lo:hi + 1].min()
    return l, u


def answers_rsm_ed(series, q, left, right, shift, eps):
    n, m = len(series), len(q)
    res = []
    for s in window_starts(left, right, shift, n, m):
        d2 = ed2(series, q, s)
        if d2 <= eps * eps * (1 + 1e-12):
            res.append((int(s), np.sqrt(d2)))
    return res


def answers_cnsm_ed(series, q, left, right, shift, eps, alpha, beta):
    n, m = len(series), len(q)
    mq, sq = znorm_stats(q)
    zq = (q - mq) / sq
    res = []
    for s in window_starts(left, right, shift, n, m):
        w = series[s - 1: s - 1 + m]
        mean, std = znorm_stats(w)
        if not (abs(mean - mq) <= beta and std / sq <= alpha and std / sq >= 1.0 / alpha):
            continue
        d = (w - mean) / std - zq
        d2 = float(np.dot(d, d))
        if d2 <= eps * eps:
            res.append((int(s), np.sqrt(d2)))
    return res
