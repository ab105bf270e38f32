#!/usr/bin/env python3
"""GPU timeline of the last phase-2 call in a rocprofv3 --kernel-trace --memory-copy-trace run: start/end of every
kernel and copy relative to the first event of the call, and the gaps between them."""
import csv, glob, sys
base = sys.argv[1]
ev = []
for f in glob.glob(base + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(base + "/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ev.sort()
# last call = events after the last gap > 2 ms ... simply take the tail starting at the last 'copy' that follows a gap > 100 us
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 8):i]) > 150_000:
        cut = i
ev = ev[cut:]
t0 = ev[0][0]
prev_end = t0
for s, e, name in ev:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {name}")
    prev_end = max(prev_end, e)
print(f"total {(prev_end - t0) / 1e3:.1f} us")
