"""Seeded synthetic inputs (thin wrappers over kvm_gen_* in libkvmatch_hip.so; host-side code, no GPU needed).

Restates the distribution of K/DataGenerator.java:80-119; see kv-match_amd/csrc/kvm_datagen.cpp."""
import numpy as np

from . import _lib

SERIES_SEED = 0x4B564D31  # "KVM1" 


def gen_series(n, seed=SERIES_SEED):
    out = np.empty(int(n), dtype=np.float64)
    _lib.check(_lib.load().kvm_gen_series(int(seed), int(n), out.ctypes.data))
    return out


def gen_intervals(n, m, target_candidates, mean_len=64.0, must_cover=(), seed=1):
    """Phase-1-shaped candidate list: sorted, disjoint, non-adjacent (left[], right[]) int32 arrays."""
    cap = int(target_candidates) + len(must_cover) + 16
    left = np.empty(cap, dtype=np.int32)
    right = np.empty(cap, dtype=np.int32)
    must = np.ascontiguousarray(must_cover, dtype=np.int32)
    k = _lib.load().kvm_gen_intervals(int(seed), int(n), int(m), int(target_candidates), float(mean_len),
                                      must.ctypes.data if must.size else None, int(must.size), left.ctypes.data,
                                      right.ctypes.data, cap)
    if k < 0:
        _lib.check(int(k))
    return left[:k].copy(), right[:k].copy()


def write_series_file(path, values):
    """files/data-{N}: flat big-endian fp64 (K/DataGenerator.java:102-112)."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    _lib.check(_lib.load().kvm_write_series_file(str(path).encode(), v.ctypes.data, v.size))


def plant_copies(series, query_offset, m, positions, scales, shifts, noise, seed=7):
    """Overwrite series[p-1 : p-1+m] with a*Q + b + noise*N(0,1) for each planted position p (1-based) so that
    verification has non-trivial answers.  Q is read before any planting."""
    rng = np.random.default_rng(seed)
    q = series[query_offset - 1: query_offset - 1 + m].copy()
    for p, a, b in zip(positions, scales, shifts):
        series[p - 1: p - 1 + m] = a * q + b + noise * rng.standard_normal(m)
    return q
