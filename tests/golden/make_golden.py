#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small fixed inputs and the oracle's outputs for the four variants.

The reference (Java) has no golden vectors and cannot run in this image, so these fixtures pin
the ORACLE's behaviour (and, through tests/test_golden.py, the HIP path's) against regressions; they are data only:
inputs (series, query offset, intervals, parameters) and expected (offsets, distances, cntCandidate).
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402
from kvmatch import datagen  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, n, seed, m, qoff, planted, C, mean_len, shift(lastSegment), params
    ("rsm_ed_a", 6000, 11, 96, 1500, [4000], 1500, 12.0, 1, dict(eps=4.0)),
    ("rsm_ed_shift", 6000, 12, 128, 700, [3000, 5000], 2500, 20.0, 3, dict(eps=6.0)),
    ("cnsm_ed_a", 6000, 13, 100, 2100, [500, 4400], 2000, 16.0, 1, dict(eps=2.5, alpha=1.5, beta=2.0)),
    ("rsm_dtw_a", 5000, 14, 64, 1200, [3300], 1200, 10.0, 1, dict(eps=3.0, rho=5)),
    ("rsm_dtw_odd", 5000, 15, 81, 900, [2500], 1000, 10.0, 2, dict(eps=3.5, rho=7)),
    ("cnsm_dtw_a", 5000, 16, 72, 2000, [600, 3900], 1500, 12.0, 1, dict(eps=2.0, rho=4, alpha=1.5, beta=2.0)),
]


def main():
    for name, n, seed, m, qoff, planted, C, mean_len, last_segment, p in CASES:
        series = datagen.gen_series(n, seed=seed)
        q = datagen.plant_copies(series, qoff, m, planted, [1.0 + 0.1 * i for i in range(len(planted))],
                                 [0.2 * i for i in range(len(planted))], 0.03, seed=seed)
        shift = (last_segment - 1) * 25
        # intervals are phase-1 output, i.e. given BEFORE the shift is subtracted: add it back
        left, right = datagen.gen_intervals(n, m, C, mean_len, must_cover=[qoff] + planted, seed=seed)
        left = (left + shift).astype(np.int32)
        right = (right + shift).astype(np.int32)
        variant = name.split("_")[0] + "_" + name.split("_")[1]
        if variant == "rsm_ed":
            off, dist, st = oracle.rsm_ed(series, q, left, right, shift, p["eps"])
        elif variant == "cnsm_ed":
            off, dist, st = oracle.cnsm_ed(series, q, left, right, shift, p["eps"], p["alpha"], p["beta"])
        elif variant == "rsm_dtw":
            off, dist, st = oracle.rsm_dtw(series, q, left, right, shift, p["eps"], p["rho"])
        else:
            off, dist, st = oracle.cnsm_dtw(series, q, left, right, shift, p["eps"], p["rho"], p["alpha"], p["beta"])
        assert qoff in off.tolist() and len(off) >= 2, (name, len(off))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), series=series, query=q, left=left, right=right,
                            last_segment=last_segment, variant=variant, params=np.array(sorted(p.items()), dtype=object),
                            exp_off=off, exp_dist=dist, exp_cnt=st["cnt_candidate"], exp_windows=st["windows"])
        print(name, "answers", len(off), "cnt", st["cnt_candidate"])


if __name__ == "__main__":
    main()
