#!/usr/bin/env python3
"""When do the waves of one launch of kvm_ed_dot_kernel start and end?  Needs an experimental build that prints, per class-A
wave, s_memrealtime at entry and exit, its XCC and HW_ID ("W slot t0 t1 xcc hwid", armed by KVM_STAMP=1, as
profiles/variants/libkvm_stamp.so: eight lines around the edd_body<8,1> call of kvm_ed_dot_kernel; not kept in the tree).  Full class-A tiles, `units` units (argv[1], default 1956), m = 8192, no split; one armed launch after the warm-up.
Prints the spread of start and end times and the mean residency per XCD."""
import os, sys, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
    import numpy as np, kvmatch
    from kvmatch import datagen
    n, m = 100_000_000, 8192
    units = int(sys.argv[2]); tiles = units * 8
    s = datagen.gen_series(n)
    q = s[n // 3: n // 3 + m].copy()
    ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0); ctx.set_option(kvmatch.OPT_TIMING, 1)
    eng = kvmatch.QueryEngine(ctx)
    starts = 1 + np.arange(tiles, dtype=np.int64) * ((n - m - 100) // tiles)
    left = starts.astype(np.int32); right = (starts + 63).astype(np.int32)
    os.environ["KVM_ED_SPLIT"] = sys.argv[3] if len(sys.argv) > 3 else "1,1"
    for _ in range(2500): eng.verify((left, right), 1, q, 10.0)
    os.environ["KVM_STAMP"] = "1"
    r = eng.verify((left, right), 1, q, 10.0)
    print("KERNEL_NS", r.stats[kvmatch.STAT_KERNEL_NS], flush=True)
    sys.exit(0)
units = sys.argv[1] if len(sys.argv) > 1 else "1956"
env = dict(os.environ); env["KVM_LIB"] = os.path.join(ROOT, "profiles", "variants", "libkvm_stamp.so")
out = subprocess.run([sys.executable, __file__, "child", units] + sys.argv[2:3], env=env, capture_output=True, text=True).stdout
import numpy as np
rows = np.array([[int(x) for x in l.split()[1:]] for l in out.splitlines() if l.startswith("W ")], dtype=np.int64)
kns = [int(l.split()[1]) for l in out.splitlines() if l.startswith("KERNEL_NS")]
t0 = rows[:, 1].min()
st, en = (rows[:, 1] - t0) / 100.0, (rows[:, 2] - t0) / 100.0   # s_memrealtime: 100 MHz -> us
print(f"{len(rows)} waves, kernel (HIP events) {kns[0] / 1e3:.1f} us; last end {en.max():.1f} us")
print("start us: " + "  ".join(f"p{p} {np.percentile(st, p):.1f}" for p in (0, 10, 50, 90, 99, 100)))
print("end   us: " + "  ".join(f"p{p} {np.percentile(en, p):.1f}" for p in (0, 1, 10, 50, 90, 100)))
print(f"mean residency {np.mean(en - st):.1f} us = {np.mean(en - st) / en.max():.3f} of the span")
for x in sorted(set(rows[:, 3].tolist())):
    k = rows[:, 3] == x
    print(f"  XCC {x}: {k.sum():4d} waves  start p50 {np.median(st[k]):6.1f}  end p50 {np.median(en[k]):6.1f}  end max {en[k].max():6.1f}  duration p50 {np.median(en[k] - st[k]):6.1f}")

# HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
hw = rows[:, 4]
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
key = rows[:, 3] * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
import collections
per_simd = collections.Counter(key.tolist())
print("waves per SIMD (SIMDs that hold any):", dict(sorted(collections.Counter(per_simd.values()).items())))
cus = collections.Counter((key // 10).tolist())
print("waves per CU:", dict(sorted(collections.Counter(cus.values()).items())), " CUs used:", len(cus))
dur = en - st
for nw in sorted(set(per_simd.values())):
    k = np.array([per_simd[x] == nw for x in key.tolist()])
    print(f"  waves on a SIMD holding {nw}: {k.sum():5d}   duration p10 {np.percentile(dur[k], 10):7.1f}  p50 {np.median(dur[k]):7.1f} us")
