// kvm_jsort.h -- host side: the permutation java.util.Arrays.sort(T[], Comparator) of JDK 8 produces, for the one call
// on the phase-2 path whose result depends on the sorting ALGORITHM: NormQueryEngineDtw sorts its z-normalised query
// with (o1, o2) -> (int)(|o2.value| - |o1.value|) (K/NormQueryEngineDtw.java:475-478), a comparator that breaks the
// Comparator contract (everything less than 1 apart compares "equal", and that is not transitive).  The order only
// fixes the summation order of the two LB_Keogh loops, but the reference's accept/reject decision at an exact tie with
// eps^2 depends on it, so the library reproduces it: TimSort as TimSort.sort(a, 0, n, c, null, 0, 0) runs it -- run
// detection with reversal of strictly descending runs, binary insertion up to minRun, the JDK 8 mergeCollapse rule,
// trimming by gallopRight / gallopLeft in mergeAt, mergeLo / mergeHi with the adaptive minGallop.
//
// Provenance.  The JDK is a dependency that is not part of the reference tree (pom.xml targets Java 8).  The ALGORITHM is
// Tim Peters' listsort (CPython's Objects/listsort.txt and listobject.c, 2002, PSF licence); java.util.TimSort is Joshua
// Bloch's port of it, first published in Android / Apache Harmony under the Apache License 2.0 and shipped in OpenJDK 7+
// (GPLv2 with the Classpath exception).  This file is a restatement written for this library from the published
// description of that algorithm and of the JDK 8 behaviour it has to reproduce -- the structure and the names of the steps
// (count_run_and_make_ascending, binary_sort, merge_collapse, gallop_left / gallop_right, merge_lo / merge_hi, MIN_MERGE,
// MIN_GALLOP) follow the JDK's unavoidably, because what has to be reproduced under a contract-breaking comparator IS
// the implementation's sequence of comparisons and moves: the order of every comparison, the JDK 8 form of the
// mergeCollapse invariant (before the 2015 fix of the stack-size bug, JDK-8072909 / de Gouw et al.), and the two places
// that throw.  No JDK source text is copied; the same algorithm is restated twice more by the test infrastructure (in C
// and in Python), and tests/test_transliteration.py holds the three to the same permutation and the
// same throw / no-throw verdict.
// Two facts a caller can rely on: a sequence with c(a[i], a[i-1]) >= 0 throughout is ONE run and comes back unchanged
// (every query whose |zQ| never rises by >= 1 between neighbours); and the sort can FAIL -- Java throws
// IllegalArgumentException("Comparison method violates its general contract!") out of mergeLo / mergeHi, which ends
// the reference's query() -- reported here as `false`.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace kvm_jsort {

constexpr int MIN_MERGE = 32, MIN_GALLOP = 7;

template <class C>
struct sorter {
    int32_t *a;
    C c;  // c(x, y): the comparator's int for elements x, y
    int min_gallop = MIN_GALLOP;
    std::vector<int32_t> tmp;
    int stack = 0;
    int run_base[64], run_len[64];

    static void copy(int32_t *dst, const int32_t *src, int n) { std::memmove(dst, src, (size_t)n * sizeof(int32_t)); }

    int count_run_and_make_ascending(int lo, int hi) {
        int run_hi = lo + 1;
        if (run_hi == hi) return 1;
        if (c(a[run_hi++], a[lo]) < 0) {
            while (run_hi < hi && c(a[run_hi], a[run_hi - 1]) < 0) run_hi++;
            for (int i = lo, j = run_hi - 1; i < j; i++, j--) {
                const int32_t t = a[i];
                a[i] = a[j];
                a[j] = t;
            }
        } else {
            while (run_hi < hi && c(a[run_hi], a[run_hi - 1]) >= 0) run_hi++;
        }
        return run_hi - lo;
    }

    void binary_sort(int lo, int hi, int start) {
        if (start == lo) start++;
        for (; start < hi; start++) {
            const int32_t pivot = a[start];
            int left = lo, right = start;
            while (left < right) {
                const int mid = (int)(((unsigned)left + (unsigned)right) >> 1);
                if (c(pivot, a[mid]) < 0) right = mid;
                else left = mid + 1;
            }
            copy(a + left + 1, a + left, start - left);
            a[left] = pivot;
        }
    }

    static int min_run_length(int n) {
        int r = 0;
        while (n >= MIN_MERGE) {
            r |= (n & 1);
            n >>= 1;
        }
        return n + r;
    }

    // position of `key` in the sorted range arr[base, base+len): leftmost (gallop_left) / rightmost (gallop_right)
    int gallop_left(int32_t key, const int32_t *arr, int base, int len, int hint) {
        int last = 0, ofs = 1;
        if (c(key, arr[base + hint]) > 0) {
            const int max_ofs = len - hint;
            while (ofs < max_ofs && c(key, arr[base + hint + ofs]) > 0) {
                last = ofs;
                ofs = (ofs << 1) + 1;
                if (ofs <= 0) ofs = max_ofs;
            }
            if (ofs > max_ofs) ofs = max_ofs;
            last += hint;
            ofs += hint;
        } else {
            const int max_ofs = hint + 1;
            while (ofs < max_ofs && c(key, arr[base + hint - ofs]) <= 0) {
                last = ofs;
                ofs = (ofs << 1) + 1;
                if (ofs <= 0) ofs = max_ofs;
            }
            if (ofs > max_ofs) ofs = max_ofs;
            const int t = last;
            last = hint - ofs;
            ofs = hint - t;
        }
        last++;
        while (last < ofs) {
            const int mid = last + ((ofs - last) >> 1);
            if (c(key, arr[base + mid]) > 0) last = mid + 1;
            else ofs = mid;
        }
        return ofs;
    }

    int gallop_right(int32_t key, const int32_t *arr, int base, int len, int hint) {
        int ofs = 1, last = 0;
        if (c(key, arr[base + hint]) < 0) {
            const int max_ofs = hint + 1;
            while (ofs < max_ofs && c(key, arr[base + hint - ofs]) < 0) {
                last = ofs;
                ofs = (ofs << 1) + 1;
                if (ofs <= 0) ofs = max_ofs;
            }
            if (ofs > max_ofs) ofs = max_ofs;
            const int t = last;
            last = hint - ofs;
            ofs = hint - t;
        } else {
            const int max_ofs = len - hint;
            while (ofs < max_ofs && c(key, arr[base + hint + ofs]) >= 0) {
                last = ofs;
                ofs = (ofs << 1) + 1;
                if (ofs <= 0) ofs = max_ofs;
            }
            if (ofs > max_ofs) ofs = max_ofs;
            last += hint;
            ofs += hint;
        }
        last++;
        while (last < ofs) {
            const int mid = last + ((ofs - last) >> 1);
            if (c(key, arr[base + mid]) < 0) ofs = mid;
            else last = mid + 1;
        }
        return ofs;
    }

    // merge of adjacent runs, the shorter one parked in tmp; false where Java throws
    bool merge_lo(int base1, int len1, int base2, int len2) {
        tmp.assign(a + base1, a + base1 + len1);
        const int32_t *t = tmp.data();
        int c1 = 0, c2 = base2, dest = base1;
        a[dest++] = a[c2++];
        if (--len2 == 0) {
            copy(a + dest, t + c1, len1);
            return true;
        }
        if (len1 == 1) {
            copy(a + dest, a + c2, len2);
            a[dest + len2] = t[c1];
            return true;
        }
        int mg = min_gallop;
        for (bool done = false; !done;) {
            int n1 = 0, n2 = 0;  // consecutive wins of run 1 / run 2
            do {
                if (c(a[c2], t[c1]) < 0) {
                    a[dest++] = a[c2++];
                    n2++;
                    n1 = 0;
                    if (--len2 == 0) { done = true; break; }
                } else {
                    a[dest++] = t[c1++];
                    n1++;
                    n2 = 0;
                    if (--len1 == 1) { done = true; break; }
                }
            } while ((n1 | n2) < mg);
            if (done) break;
            do {
                n1 = gallop_right(a[c2], t, c1, len1, 0);
                if (n1 != 0) {
                    copy(a + dest, t + c1, n1);
                    dest += n1;
                    c1 += n1;
                    len1 -= n1;
                    if (len1 <= 1) { done = true; break; }
                }
                a[dest++] = a[c2++];
                if (--len2 == 0) { done = true; break; }
                n2 = gallop_left(t[c1], a, c2, len2, 0);
                if (n2 != 0) {
                    copy(a + dest, a + c2, n2);
                    dest += n2;
                    c2 += n2;
                    len2 -= n2;
                    if (len2 == 0) { done = true; break; }
                }
                a[dest++] = t[c1++];
                if (--len1 == 1) { done = true; break; }
                mg--;
            } while ((n1 >= MIN_GALLOP) | (n2 >= MIN_GALLOP));
            if (done) break;
            if (mg < 0) mg = 0;
            mg += 2;
        }
        min_gallop = mg < 1 ? 1 : mg;
        if (len1 == 1) {
            copy(a + dest, a + c2, len2);
            a[dest + len2] = t[c1];
        } else if (len1 == 0) {
            return false;
        } else {
            copy(a + dest, t + c1, len1);
        }
        return true;
    }

    bool merge_hi(int base1, int len1, int base2, int len2) {
        tmp.assign(a + base2, a + base2 + len2);
        const int32_t *t = tmp.data();
        int c1 = base1 + len1 - 1, c2 = len2 - 1, dest = base2 + len2 - 1;
        a[dest--] = a[c1--];
        if (--len1 == 0) {
            copy(a + dest - (len2 - 1), t, len2);
            return true;
        }
        if (len2 == 1) {
            dest -= len1;
            c1 -= len1;
            copy(a + dest + 1, a + c1 + 1, len1);
            a[dest] = t[c2];
            return true;
        }
        int mg = min_gallop;
        for (bool done = false; !done;) {
            int n1 = 0, n2 = 0;
            do {
                if (c(t[c2], a[c1]) < 0) {
                    a[dest--] = a[c1--];
                    n1++;
                    n2 = 0;
                    if (--len1 == 0) { done = true; break; }
                } else {
                    a[dest--] = t[c2--];
                    n2++;
                    n1 = 0;
                    if (--len2 == 1) { done = true; break; }
                }
            } while ((n1 | n2) < mg);
            if (done) break;
            do {
                n1 = len1 - gallop_right(t[c2], a, base1, len1, len1 - 1);
                if (n1 != 0) {
                    dest -= n1;
                    c1 -= n1;
                    len1 -= n1;
                    copy(a + dest + 1, a + c1 + 1, n1);
                    if (len1 == 0) { done = true; break; }
                }
                a[dest--] = t[c2--];
                if (--len2 == 1) { done = true; break; }
                n2 = len2 - gallop_left(a[c1], t, 0, len2, len2 - 1);
                if (n2 != 0) {
                    dest -= n2;
                    c2 -= n2;
                    len2 -= n2;
                    copy(a + dest + 1, t + c2 + 1, n2);
                    if (len2 <= 1) { done = true; break; }
                }
                a[dest--] = a[c1--];
                if (--len1 == 0) { done = true; break; }
                mg--;
            } while ((n1 >= MIN_GALLOP) | (n2 >= MIN_GALLOP));
            if (done) break;
            if (mg < 0) mg = 0;
            mg += 2;
        }
        min_gallop = mg < 1 ? 1 : mg;
        if (len2 == 1) {
            dest -= len1;
            c1 -= len1;
            copy(a + dest + 1, a + c1 + 1, len1);
            a[dest] = t[c2];
        } else if (len2 == 0) {
            return false;
        } else {
            copy(a + dest - (len2 - 1), t, len2);
        }
        return true;
    }

    bool merge_at(int i) {
        int base1 = run_base[i], len1 = run_len[i];
        const int base2 = run_base[i + 1];
        int len2 = run_len[i + 1];
        run_len[i] = len1 + len2;
        if (i == stack - 3) {
            run_base[i + 1] = run_base[i + 2];
            run_len[i + 1] = run_len[i + 2];
        }
        stack--;
        const int k = gallop_right(a[base2], a, base1, len1, 0);
        base1 += k;
        len1 -= k;
        if (len1 == 0) return true;
        len2 = gallop_left(a[base1 + len1 - 1], a, base2, len2, len2 - 1);
        if (len2 == 0) return true;
        return len1 <= len2 ? merge_lo(base1, len1, base2, len2) : merge_hi(base1, len1, base2, len2);
    }

    bool merge_collapse() {
        while (stack > 1) {
            int n = stack - 2;
            if (n > 0 && run_len[n - 1] <= run_len[n] + run_len[n + 1]) {
                if (run_len[n - 1] < run_len[n + 1]) n--;
                if (!merge_at(n)) return false;
            } else if (run_len[n] <= run_len[n + 1]) {
                if (!merge_at(n)) return false;
            } else {
                break;
            }
        }
        return true;
    }

    bool merge_force_collapse() {
        while (stack > 1) {
            int n = stack - 2;
            if (n > 0 && run_len[n - 1] < run_len[n + 1]) n--;
            if (!merge_at(n)) return false;
        }
        return true;
    }

    bool sort(int n) {
        int lo = 0, remaining = n;
        if (remaining < 2) return true;
        if (remaining < MIN_MERGE) {
            const int init = count_run_and_make_ascending(lo, n);
            binary_sort(lo, n, lo + init);
            return true;
        }
        const int min_run = min_run_length(remaining);
        do {
            int len = count_run_and_make_ascending(lo, n);
            if (len < min_run) {
                const int force = remaining <= min_run ? remaining : min_run;
                binary_sort(lo, lo + force, lo + len);
                len = force;
            }
            if (stack == 64) return false;  // (cannot happen below 2^31 elements: the depth is logarithmic; Java's own
                                            //  stack holds 49 entries and would throw ArrayIndexOutOfBoundsException)
            run_base[stack] = lo;
            run_len[stack] = len;
            stack++;
            if (!merge_collapse()) return false;
            lo += len;
            remaining -= len;
        } while (remaining != 0);
        return merge_force_collapse();
    }
};

// Sorts the n elements of `a` in place like Arrays.sort(a, c); false where Java throws IllegalArgumentException.
template <class C>
inline bool java_arrays_sort(int32_t *a, int n, C c) {
    sorter<C> s{a, c};
    return s.sort(n);
}

// (int) of a double: toward zero, NaN -> 0, saturating
inline int java_int(double v) {
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return -2147483647 - 1;
    return (int)v;
}

// order[] of NormQueryEngineDtw for the z-normalised query zq[0, m): false where the reference's sort throws
inline bool cnsm_dtw_order(const double *zq, int m, int32_t *order) {
    for (int i = 0; i < m; i++) order[i] = i;
    return java_arrays_sort(order, m, [zq](int32_t o1, int32_t o2) {
        const double a1 = zq[o1] < 0 ? -zq[o1] : zq[o1], a2 = zq[o2] < 0 ? -zq[o2] : zq[o2];
        return java_int(a2 - a1);
    });
}

}  // namespace kvm_jsort
