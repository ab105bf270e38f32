#!/usr/bin/env python3
"""Fit of the two constants of the 16-per-lane form of class A in the ED pruner's launch model (kvm_ed_launch_model.h:
a part costs a16 + b16 ceil(chunks / f) us), everything else in the model as compiled in, to profiles/ed_a16_sweep.py's
log (per shape: the traced unit counts of both forms, then the kernel times of the nine split pairs per form and of the
launcher's own choice).  Prints the constants and how far the launcher's choice of (form, pair) is behind the best measured
of the eighteen, with the constants in the header and with the fitted ones.

    python profiles/fit_ed_a16.py gpurun_out/r05_a16_sweep.log"""
import math, re, sys
import numpy as np
from scipy.optimize import least_squares
SLOTS = 2048.0
KX = [1.0, 0.75, 0.55, 0.5, 0.25, 0.0]; KV = [1.0, 1.093, 1.149, 1.687, 1.695, 2.240]
PAIRS = [(a, b) for a in (1, 2, 4) for b in (1, 2, 4)]
W8 = (5.188, 4.535); WC = (6.360, 3.899); WB = (3.951, 3.719)
def speed(x):
    if x >= 1.0: return 1.0
    for i in range(1, 6):
        if x >= KX[i]: return KV[i - 1] + (KV[i] - KV[i - 1]) * (KX[i - 1] - x) / (KX[i - 1] - KX[i])
    return KV[5]
def makespan(cnt, w):
    cnt = list(cnt); act = []; qi = 0; free = SLOTS; t = 0.0
    for _ in range(100000):
        while free > 0 and qi < 3:
            if cnt[qi] <= 0: qi += 1; continue
            take = min(free, cnt[qi]); act.append([take, w[qi]]); free -= take; cnt[qi] -= take
        if not act: return t
        n = sum(a[0] for a in act); rem = min(a[1] for a in act)
        t += rem / speed(n / SLOTS)
        nxt = []
        for a in act:
            a[1] -= rem
            if a[1] <= 1e-9: free += a[0]
            else: nxt.append(a)
        act = nxt
    return t
def model(m, uA, uB, uC, fA, fB, wa):
    ch = lambda f: math.ceil(math.ceil(m / 128) / f)
    w = [wa[0] + wa[1] * ch(fA), WC[0] + WC[1] * ch(fB), WB[0] + WB[1] * ch(fB)]
    parts = lambda u, f: 4.0 * math.ceil(u * f / 4.0)
    return makespan([parts(uA, fA), parts(uC, fB), parts(uB, fB)], w)
rows = []; units = {}
for line in open(sys.argv[1]):
    mm = re.search(r"ed_dot m (\d+) unitsA (\d+) \((\d+) per lane\) unitsB (\d+) unitsC (\d+)", line)
    if mm: units[int(mm.group(3))] = tuple(int(mm.group(i)) for i in (1, 2, 4, 5)); continue
    if line.startswith("m ") and 8 in units and 16 in units:
        f8, f16 = re.search(r"form 8: (.*?) \| form 16: (.*?) \| auto ([0-9.]+)", line).group(1, 2)
        t8 = [float(x) * 1e3 for x in re.findall(r"\d,\d ([0-9.]+)", f8)]; t16 = [float(x) * 1e3 for x in re.findall(r"\d,\d ([0-9.]+)", f16)]
        auto = float(re.search(r"auto ([0-9.]+)", line).group(1)) * 1e3
        rows.append((units[8], units[16], t8, t16, auto)); units = {}
print(len(rows), "shapes")
def resid(th):
    out = []
    for u8, u16, t8, t16, _ in rows:
        m, uA, uB, uC = u16
        for (a, b), tv in zip(PAIRS, t16): out.append(math.log(max(model(m, uA, uB, uC, a, b, th) + 6.0, 1e-3)) - math.log(tv))
    return out
def behind(th):
    r = []
    for u8, u16, t8, t16, _ in rows:
        best, pick = 1e300, None
        for form, (m, uA, uB, uC), wa in ((8, u8, W8), (16, u16, th)):
            if uA + uB + uC >= 16 * SLOTS:
                if form == 16: pick = (16, 0)
                continue
            b2, p2 = 1e300, 0
            for i, (a, b) in enumerate(PAIRS):
                tv = model(m, uA, uB, uC, a, b, wa)
                if tv < b2 * (1 - 1e-3): b2, p2 = tv, i
            if b2 < best: best, pick = b2, (form, p2)
        t = (t8 if pick[0] == 8 else t16)[pick[1]]
        r.append(t / min(min(t8), min(t16)))
    return math.exp(np.mean(np.log(r))), max(r)
th0 = [9.0, 7.75]
print("header constants", th0, "behind the best measured (geo mean, worst): %.3f %.3f" % behind(th0))
sol = least_squares(resid, th0, bounds=([0.0, 1.0], [60.0, 20.0]))
print("fitted a16 %.3f b16 %.3f  rms log error %.3f" % (sol.x[0], sol.x[1], math.sqrt(np.mean(np.square(sol.fun)))))
print("fitted: behind the best measured (geo mean, worst): %.3f %.3f" % behind(sol.x))
print("measured launcher (auto) behind the best: geo mean %.3f worst %.3f" % (math.exp(np.mean([math.log(r[4] / min(min(r[2]), min(r[3]))) for r in rows])), max(r[4] / min(min(r[2]), min(r[3])) for r in rows)))
