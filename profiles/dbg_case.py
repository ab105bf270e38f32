"""Debug helper: one case of the adversarial generator (SEED, RD, KIND in the environment), cNSM-ED on its full list."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kvmatch
from kvmatch import datagen
from oracle import oracle
import adversarial
seed, want_rd, want_kind = int(os.environ.get("SEED", "11")), int(os.environ.get("RD", "2")), os.environ.get("KIND", "tiny")
for rd, kind, n, m, rho, s, q, qoff, pl, lists, eps_tie, eps_generic in adversarial.cases(datagen, want_rd + 1, seed):
    if (rd, kind) == (want_rd, want_kind):
        break
print("case", rd, kind, "n", n, "m", m, "qoff", qoff, "planted", pl, "eps_tie", eps_tie)
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1); oracle.set_lenient_sort(True)  # compare the arithmetic also where the reference's sort throws
name, L, R = lists[2]
for alpha, beta in ((1.5, 0.5),):
    a = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps_tie, alpha, beta)
    o = oracle.cnsm_ed(s, q, L, R, 0, eps_tie, alpha, beta)
    print("got", len(a), "want", len(o[0]), "lib stats", a.stats, "oracle", {k: o[2][k] for k in ("pass_filter", "windows")})
    missing = sorted(set(o[0].tolist()) - set(a.offsets.tolist()))
    extra = sorted(set(a.offsets.tolist()) - set(o[0].tolist()))
    od = dict(zip(o[0].tolist(), o[1].tolist()))
    print("missing", [(x, od[x]) for x in missing], "extra", extra)
    mq, sq = oracle.query_stats(q)
    for x in missing:
        w = s[x - 1: x - 1 + m]
        print(" offset", x, "window min/max", w.min(), w.max(), "np mean/std", w.mean(), w.std(), "dist to planted", [x - p for p in pl], "qoff", x - qoff)
    print("meanQ stdQ", mq, sq)
print("---- by ED kernel / list")
for kern in (1, 2, 3, 4):
    ctx.set_option(kvmatch.OPT_ED_KERNEL, kern)
    for (l0, r0) in ((1, n - m + 1), (36700, 36900), (36785, 36795), (36790, 36792)):
        Lx, Rx = np.array([l0], dtype=np.int32), np.array([r0], dtype=np.int32)
        a = kvmatch.NormQueryEngine(ctx).verify((Lx, Rx), 1, q, eps_tie, 1.5, 0.5)
        o = oracle.cnsm_ed(s, q, Lx, Rx, 0, eps_tie, 1.5, 0.5)
        miss = sorted(set(o[0].tolist()) - set(a.offsets.tolist()))
        print("kernel", kern, "list", (l0, r0), "got", len(a), "want", len(o[0]), "missing", miss[:6], "stats", a.stats[:2], a.stats[6], "oracle pass_filter", o[2]["pass_filter"])
