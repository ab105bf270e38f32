"""Debug helper: one case of the adversarial generator (SEED, RD, KIND, LIST, EPS, ALPHA, BETA in the environment), cNSM-ED."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kvmatch
from kvmatch import datagen
from oracle import oracle
import adversarial
E = os.environ.get
seed, want_rd, want_kind = int(E("SEED", "23")), int(E("RD", "0")), E("KIND", "tiny")
for rd, kind, n, m, rho, s, q, qoff, pl, lists, eps_tie, eps_generic in adversarial.cases(datagen, want_rd + 1, seed):
    if (rd, kind) == (want_rd, want_kind):
        break
li = int(E("LIST", "2"))
name, L, R = lists[li]
eps = float(E("EPS", str(eps_generic[li]))); alpha = float(E("ALPHA", "1e9")); beta = float(E("BETA", "1e9"))
print("case", rd, kind, "n", n, "m", m, "qoff", qoff, "planted", pl, "list", name, len(L), "eps", eps, "eps_tie", eps_tie, "eps_generic", eps_generic)
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
o = oracle.cnsm_ed(s, q, L, R, 0, eps, alpha, beta)
od = dict(zip(o[0].tolist(), o[1].tolist()))
for kern in (3, 1, 2, 4):
    for ab in (1, 0):
        ctx.set_option(kvmatch.OPT_ED_KERNEL, kern); ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
        a = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, alpha, beta)
        missing = sorted(set(o[0].tolist()) - set(a.offsets.tolist()))
        print("kernel", kern, "ab", ab, "got", len(a), "want", len(o[0]), "stats", a.stats, "missing head", [(x, round(od[x], 4)) for x in missing[:5]], "dist range of missing", (min(od[x] for x in missing), max(od[x] for x in missing)) if missing else None)
ctx.set_option(kvmatch.OPT_ED_KERNEL, 0); ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
mq, sq = oracle.query_stats(q)
print("meanQ stdQ", mq, sq, "series scale", np.abs(s).max())
a = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, alpha, beta)
missing = sorted(set(o[0].tolist()) - set(a.offsets.tolist()))
import math
for x in missing[:6]:
    w = s[x - 1:x - 1 + m]
    # the reference's running-sum statistics need the interval start; show plain and long-double versions instead
    ex = float(np.sum(w)); ex2 = float(np.sum(w * w)); mean = ex / m; var = ex2 / m - mean * mean
    wl = w.astype(np.longdouble) * 1e158
    print(" offset", x, "ref dist", od[x], "plain var", var, "sd", math.sqrt(var) if var > 0 else None, "true sd*1e158", float(wl.std()), "ratio", (math.sqrt(var) * 1e158 / float(wl.std())) if var > 0 else None)
print("---- all distances (eps = 1e9) library vs oracle on the same list")
a = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 1e9, alpha, beta)
o = oracle.cnsm_ed(s, q, L, R, 0, 1e9, alpha, beta)
da = dict(zip(a.offsets.tolist(), a.distances.tolist())); do = dict(zip(o[0].tolist(), o[1].tolist()))
print("answers", len(da), len(do))
common = sorted(set(da) & set(do))
diff = [(x, da[x], do[x]) for x in common if da[x] != do[x]]
print("common", len(common), "differing", len(diff))
if diff:
    rel = np.array([abs(a_ - b_) / max(b_, 1e-300) for _, a_, b_ in diff])
    print("rel diff max", rel.max(), "median", np.median(rel), "first", diff[:5])
    xs = np.array([x for x, _, _ in diff]); print("offset range of differing", xs.min(), xs.max(), "count by 5000-bucket", np.bincount(xs // 5000))
