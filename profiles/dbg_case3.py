"""Debug helper: run the adversarial harness on ONE case (SEED, RD, KIND), printing every verification before it starts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kvmatch
from kvmatch import datagen
from oracle import oracle
import adversarial
E = os.environ.get
seed, want_rd, want_kind = int(E("SEED", "29")), int(E("RD", "1")), E("KIND", "halfint")
for rd, kind, n, m, rho, s, q, qoff, pl, lists, eps_tie, eps_generic in adversarial.cases(datagen, want_rd + 1, seed):
    if (rd, kind) == (want_rd, want_kind):
        break
print("case", rd, kind, "n", n, "m", m, "rho", rho, "qoff", qoff, "eps_tie", eps_tie, "eps_generic", eps_generic, [(nm, len(L)) for nm, L, R in lists], flush=True)
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1); oracle.set_lenient_sort(True)
side = E("SIDE", "lib")
for li, (name, L, R) in enumerate(lists):
    for eps in (0.0, eps_tie, eps_generic[li]):
        for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (1.5, 0.5), (1e9, 1e9)):
            for v in ("rsm_ed", "cnsm_ed", "rsm_dtw", "cnsm_dtw"):
                if v.startswith("rsm") and not (alpha == 1.0 and beta == 0.0): continue
                if v.endswith("dtw") and name == "full" and n >= 60_000: continue
                t = time.perf_counter()
                print(f"{side} {v} list={name} eps={eps:.4f} a={alpha} b={beta} ...", end="", flush=True)
                if side == "lib":
                    r = {"rsm_ed": lambda: kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps), "cnsm_ed": lambda: kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, alpha, beta),
                         "rsm_dtw": lambda: kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps, int(rho)), "cnsm_dtw": lambda: kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho, alpha, beta, resolved=True)}[v]()
                    print(f" {time.perf_counter() - t:.2f} s, {len(r)} answers, stats {r.stats}", flush=True)
                else:
                    r = {"rsm_ed": lambda: oracle.rsm_ed(s, q, L, R, 0, eps), "cnsm_ed": lambda: oracle.cnsm_ed(s, q, L, R, 0, eps, alpha, beta),
                         "rsm_dtw": lambda: oracle.rsm_dtw(s, q, L, R, 0, eps, int(rho)), "cnsm_dtw": lambda: oracle.cnsm_dtw(s, q, L, R, 0, eps, rho, alpha, beta)}[v]()
                    print(f" {time.perf_counter() - t:.2f} s, {len(r[0])} answers", flush=True)
