#!/usr/bin/env python3
"""Kernel overlap in a rocprofv3 --kernel-trace run of a multi-stream workload: per kernel average duration and count, the
union of busy time, and how much of the busy time has 1, 2, 3 ... kernels running.   overlap.py <dir> [skip_first_fraction]"""
import csv, glob, sys, collections
ev = []
for f in glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], r.get("Queue_Id", "")))
ev.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = ev[int(len(ev) * skip):]
by = collections.defaultdict(list)
for s, e, n, q in ev: by[n].append(e - s)
span = max(e for _, e, _, _ in ev) - ev[0][0]
print(f"events {len(ev)}  span {span/1e3:.0f} us  queues {len(set(q for *_, q in ev))}")
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:48s} calls {len(v):6d}  avg {sum(v)/len(v)/1e3:8.1f} us  total {sum(v)/1e3:10.0f} us  ({sum(v)/span:.2f} of the span)")
pts = []
for s, e, _, _ in ev: pts += [(s, 1), (e, -1)]
pts.sort()
lvl, last, hist = 0, pts[0][0], collections.Counter()
for t, d in pts:
    hist[lvl] += t - last; last = t; lvl += d
tot = sum(hist.values())
print("  concurrency: " + "  ".join(f"{k}: {v/tot:.2f}" for k, v in sorted(hist.items())))
