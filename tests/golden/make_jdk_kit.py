#!/usr/bin/env python3
"""Writes a directory a maintainer WITH a JDK can run the reference on, to pin the oracle (and through it this library) to
the Java engines themselves -- the one check nothing in this image can make (no JDK, no Maven).

    python tests/golden/make_jdk_kit.py <out_dir>          # needs numpy and gcc (the oracle); no GPU, no hipcc

Per golden fixture tests/golden/<case>.npz it writes <out_dir>/<case>/ with everything the reference's file-backed engines read
and what they should print:

    files/data-<n>            the series, big-endian doubles (TimeSeriesFileOperator's format, K/operator/file/TimeSeriesFileOperator.java:50)
    files/index-<n>-<Wu>      the five index files, Wu = 25 50 100 200 400, as IndexBuilder + IndexFileOperator.writeAll lay them
                              out (K/IndexBuilder.java:186-347, K/operator/file/IndexFileOperator.java:45) -- the oracle's bytes, which the library's
                              kvm_index_build reproduces byte for byte (tests/test_index_gpu.py)
    expected/index-<n>-<Wu>   the same bytes again, to `cmp` against what the reference's own IndexBuilder writes into files/
    stdin.txt                 the REPL input of the case's engine (README.md:57-70 of the reference): n, then one
                              Offset / Length / Epsilon [/ Rho] [/ Alpha / Beta] group per query, then Offset = -1
    expected.json             per query: Best offset, its distance, #answers -- the ORACLE's answers over ALL offsets
                              [1, n - m + 1] (the engines lose no answer in phase 1, so their answers are the full scan's)
    run.sh                    the java command, and the grep that extracts the lines to compare

The queries of a case: the fixture's own epsilon, then epsilons set EXACTLY at the distances of individual answers (the decimal
text Java's Double.toString and Python's repr both print round-trips), where `dist <= eps * eps` is decided by one rounding --
the sharpest cheap probe of "same arithmetic in the same order".

Only data and a recipe: nothing of the reference is copied, nothing Java travels anywhere.  The oracle is used here as
what it is -- test infrastructure."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402

WU = (25, 50, 100, 200, 400)
ENGINE = {"rsm_ed": "QueryEngine", "cnsm_ed": "NormQueryEngine", "rsm_dtw": "QueryEngineDtw", "cnsm_dtw": "NormQueryEngineDtw"}


def full_scan(variant, series, q, p, eps):
    n, m = len(series), len(q)
    left, right = np.array([1], np.int32), np.array([n - m + 1], np.int32)
    if variant == "rsm_ed":
        return oracle.rsm_ed(series, q, left, right, 0, eps)
    if variant == "cnsm_ed":
        return oracle.cnsm_ed(series, q, left, right, 0, eps, p["alpha"], p["beta"])
    if variant == "rsm_dtw":
        return oracle.rsm_dtw(series, q, left, right, 0, eps, int(p["rho"]))
    return oracle.cnsm_dtw(series, q, left, right, 0, eps, int(p["rho"]), p["alpha"], p["beta"])


def write_case(out_dir, name, only_index=None):
    z = np.load(os.path.join(HERE, name + ".npz"), allow_pickle=True)
    series, q, variant = z["series"], z["query"], str(z["variant"])
    p = {k: float(v) for k, v in z["params"]}
    n, m = len(series), len(q)
    hits = np.nonzero(series[: n - m + 1] == q[0])[0]
    qoff = next(int(h) + 1 for h in hits if np.array_equal(series[h: h + m], q))   # the query IS a slice of the series
    d = os.path.join(out_dir, name)
    os.makedirs(os.path.join(d, "files"), exist_ok=True)
    os.makedirs(os.path.join(d, "expected"), exist_ok=True)
    with open(os.path.join(d, "files", f"data-{n}"), "wb") as f:
        f.write(oracle.encode_be64(series))
    for w in (WU if only_index is None else only_index):
        raw, _ = oracle.index_file(series, w)
        for sub in ("files", "expected"):
            with open(os.path.join(d, sub, f"index-{n}-{w}"), "wb") as f:
                f.write(raw)
    # the queries: the fixture's epsilon, then epsilons exactly at the distances of (up to three) individual answers
    off, dist, _ = full_scan(variant, series, q, p, p["eps"])
    eps_list = [p["eps"]]
    for dv in sorted(set(float(x) for x in dist if x > 0.0))[:3]:
        eps_list.append(float(repr(dv)))
    queries, lines = [], [str(n)]
    for eps in eps_list:
        off, dist, _ = full_scan(variant, series, q, p, eps)
        order = np.lexsort((off, dist))     # answers.sort(Comparator.comparing(Pair::getSecond)) on an offset-ascending list: stable
        queries.append({"offset": qoff, "length": m, "epsilon": repr(eps), "best_offset": int(off[order[0]]) if len(off) else None,
                        "best_distance": repr(float(dist[order[0]])) if len(off) else None, "answers": int(len(off))})
        lines += [str(qoff), str(m), repr(eps)]
        if variant.endswith("dtw"):
            lines.append(str(int(p["rho"])))          # > 1: taken as points (K/QueryEngineDtw.java:135-140); the fixtures' rho are 4 .. 7
        if variant.startswith("cnsm"):
            lines += [repr(p["alpha"]), repr(p["beta"])]
    lines.append("-1")
    with open(os.path.join(d, "stdin.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(d, "expected.json"), "w") as f:
        json.dump({"case": name, "engine": "cn.edu.fudan.dsm.kvmatch." + ENGINE[variant], "n": n, "queries": queries}, f, indent=1)
    with open(os.path.join(d, "run.sh"), "w") as f:
        f.write("#!/bin/sh\n# run from this directory; JAR = path to KV-match-1.0-SNAPSHOT.jar (mvn package in the reference)\n"
                f"java -cp \"${{JAR:?set JAR}}\" cn.edu.fudan.dsm.kvmatch.{ENGINE[variant]} < stdin.txt 2>&1 | grep -E 'Best:|#answers:'\n"
                "# compare with expected.json: one 'Best: <offset>, distance: <d>' and one '... #answers: <k>' line per query, in order\n")
    os.chmod(os.path.join(d, "run.sh"), 0o755)
    return d, queries


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    out = sys.argv[1]
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            d, queries = write_case(out, fn[:-4])
            print(d, [(qq["epsilon"], qq["answers"]) for qq in queries])


if __name__ == "__main__":
    main()
