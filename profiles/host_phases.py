#!/usr/bin/env python3
"""Where a latency-bound call spends its time on the HOST: the library's own phase times (KVM_TRACE=1, stderr) averaged over
queries of the bench list, RSM-ED with the reference's early abandoning and cNSM-ED, one query in flight.
    python profiles/host_phases.py 2> gpurun_out/trace.err ; the script re-reads its own stderr file when given as argv[1]"""
import os, sys, re, collections
if len(sys.argv) > 1:
    acc = collections.defaultdict(list); sect = None
    for line in open(sys.argv[1]):
        if line.startswith("== "):
            sect = None if line.strip().endswith(" end") else line.strip()
            continue
        m = re.match(r"\[kvm\] (.+?)\s+([0-9.]+) us", line)
        if m and sect: acc[(sect, m.group(1).strip())].append(float(m.group(2)))
    last = None
    for (sect, lab), v in acc.items():
        if sect != last: print(sect); last = sect
        v = v[len(v) // 5:]
        print(f"   {lab:26s} {sum(v) / len(v):8.1f} us  (n={len(v)})")
    sys.exit(0)
os.environ["KVM_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd"))
import numpy as np, kvmatch, time
from kvmatch import datagen
n, m = 100_000_000, 8192
s = datagen.gen_series(n)
q = s[n // 3: n // 3 + m].copy()
left, right = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[n // 3 + 1], seed=1)
ctx = kvmatch.Phase2Context(0); ctx.load_series(s)
for name, eng, args in (("rsm_ed abandon", kvmatch.QueryEngine(ctx), (10.0,)), ("cnsm_ed abandon", kvmatch.NormQueryEngine(ctx), (10.0, 2.0, 10.0))):
    for _ in range(300): eng.verify((left, right), 1, q, *args)
    sys.stderr.write(f"== {name}\n"); sys.stderr.flush()
    t0 = time.perf_counter()
    for _ in range(100): eng.verify((left, right), 1, q, *args)
    dt = (time.perf_counter() - t0) / 100
    sys.stderr.write(f"== {name} end\n"); sys.stderr.flush()
    print(f"{name}: {dt * 1e6:.1f} us per call (one in flight, tracing on)")
