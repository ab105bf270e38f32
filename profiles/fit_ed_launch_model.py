#!/usr/bin/env python3
"""Refit of the ED pruner's launch model (kv-match_amd/csrc/kvm_ed_launch_model.h) to split sweeps taken with SWEEP_UNITS=1
(profiles/ed_split_sweep.py: per list shape one traced call -- the launcher's unit counts -- and the kernel time of the nine
split pairs).  The model: a part costs a + b ceil(chunks / f) us on a full chip; parts start in launch order A, C, B on 2048
wave slots; resident waves advance at edd_speed(fraction of slots occupied).  Fits the six (a, b), the speed knots and a launch
constant by least squares on log time; prints the constants and how far the pair the refitted model picks is behind the best
measured pair (geometric mean / worst), next to the same figure for the constants in the header.

    python profiles/fit_ed_launch_model.py gpurun_out/r03_split_sweep_a.log gpurun_out/r03_split_sweep_b.log"""
import math, re, sys
import numpy as np
from scipy.optimize import least_squares

SLOTS = 2048.0
KX = [1.0, 0.75, 0.55, 0.5, 0.25, 0.0]
PAIRS = [(a, b) for a in (1, 2, 4) for b in (1, 2, 4)]

def speed(x, kv):
    if x >= 1.0: return 1.0
    for i in range(1, 6):
        if x >= KX[i]: return kv[i - 1] + (kv[i] - kv[i - 1]) * (KX[i - 1] - x) / (KX[i - 1] - KX[i])
    return kv[5]

def makespan(cnt, w, kv):
    cnt = list(cnt); act = []; qi = 0; free = SLOTS; t = 0.0
    for _ in range(100000):
        while free > 0 and qi < 3:
            if cnt[qi] <= 0: qi += 1; continue
            take = min(free, cnt[qi]); act.append([take, w[qi]]); free -= take; cnt[qi] -= take
        if not act: return t
        n = sum(a[0] for a in act); rem = min(a[1] for a in act)
        t += rem / speed(n / SLOTS, kv)
        nxt = []
        for a in act:
            a[1] -= rem
            if a[1] <= 1e-9: free += a[0]
            else: nxt.append(a)
        act = nxt
    return t

def model(m, uA, uB, uC, fA, fB, th, jc):
    ch = lambda f, j: math.ceil(math.ceil(m / j) / f)
    w = [th[0] + th[1] * ch(fA, jc[0]), th[2] + th[3] * ch(fB, jc[1]), th[4] + th[5] * ch(fB, jc[2])]   # A, C, B
    parts = lambda u, f: 4.0 * math.ceil(u * f / 4.0)
    kv = [1.0] + list(th[6:11])
    return makespan([parts(uA, fA), parts(uC, fB), parts(uB, fB)], w, kv) + th[11]

rows = []
for fn in sys.argv[1:]:
    units = None
    for line in open(fn):
        mm = re.search(r"ed_dot m (\d+) unitsA (\d+) unitsB (\d+) unitsC (\d+)", line)
        if mm: units = tuple(int(x) for x in mm.groups()); continue
        if line.startswith("m ") and units:
            t = [float(x) for x in re.findall(r"\d,\d ([0-9.]+)", line)]
            auto = float(re.search(r"auto ([0-9.]+)", line).group(1))
            rows.append((units, [x * 1e3 for x in t], auto * 1e3)); units = None
print(len(rows), "shapes")

OLD = [9.222, 5.105, 7.968, 2.879, 5.979, 4.543, 1.073, 1.094, 1.664, 1.765, 2.187, 0.0]
OLD_JC = (128, 80, 160)
NEW_JC = (128, 128, 128)

def behind(th, jc):
    r = []
    for (m, uA, uB, uC), t, _ in rows:
        pick, best = 0, 1e300   # the launcher's rule: pairs in the order (1,1) (1,2) ... (4,4), near ties go to the coarser split
        if uA + uB + uC < 16 * SLOTS:
            for i, (a, b) in enumerate(PAIRS):
                tv = model(m, uA, uB, uC, a, b, th, jc)
                if tv < best * (1.0 - 1e-3): best, pick = tv, i
        r.append(t[pick] / min(t))
    return math.exp(np.mean(np.log(r))), max(r)

def resid(th):
    out = []
    for (m, uA, uB, uC), t, _ in rows:
        for (a, b), tv in zip(PAIRS, t): out.append(math.log(max(model(m, uA, uB, uC, a, b, th, NEW_JC), 1e-3)) - math.log(tv))
    return out

print("header constants: pick behind best  geo %.4f worst %.3f" % behind(OLD, OLD_JC))
print("launcher as run : behind best        geo %.4f worst %.3f" % (math.exp(np.mean([math.log(a / min(t)) for _, t, a in rows])), max(a / min(t) for _, t, a in rows)))
x0 = np.array([2.0, 5.4, 2.0, 4.5, 2.0, 3.5, 1.073, 1.094, 1.664, 1.765, 2.187, 5.0])
lo = [0, 1, 0, 1, 0, 1, 1, 1, 1, 1, 1, 0]; hi = [30, 12, 30, 12, 30, 12, 3, 3, 4, 4, 5, 40]
fit = least_squares(resid, x0, bounds=(lo, hi), x_scale=1.0, diff_step=1e-3)
th = fit.x
print("rms log error %.3f" % math.sqrt(np.mean(np.square(fit.fun))))
print("A: %.3f + %.3f ch   C: %.3f + %.3f ch   B: %.3f + %.3f ch   kv %s   launch %.2f" % (th[0], th[1], th[2], th[3], th[4], th[5], np.round(th[6:11], 3), th[11]))
print("refitted        : pick behind best  geo %.4f worst %.3f" % behind(th, NEW_JC))
