#!/usr/bin/env python3
"""round 5: what the (alpha, beta) filter leaves of the bench's cNSM-ED lists -- per query the candidates that pass, the
trimmed class-A / class-B tiles they are in, and the share of those tiles' lane slots they fill (KVM_TRACE lines of single calls)."""
import os, re, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, KVM_TRACE="1")
cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cnsm_ed", "--steps", "1", "--warmup", "0", "--queries-per-step", "10",
                     "--in-flight", "1", "--extras", "0", "--cpu-baseline", "0", "--sub-workloads", "0"], env=env, capture_output=True, text=True)
seen = {}
for line in cp.stderr.splitlines():
    m = re.search(r"tiles (\d+)\+(\d+)\+(\d+)\s+retiled A (\d+) B (\d+) C (\d+)\s+pruner survivors (\d+)", line)
    if m:
        key = tuple(int(x) for x in m.groups()[:6])
        seen[key] = seen.get(key, 0) + 1
d = json.loads(cp.stdout.strip().splitlines()[-1])
pl = d["roofline"].get("per_launch", [])
print("per launch (candidates evaluated, kernel ms, frac):", [(p["candidates_evaluated"], round(p["kernel_ms"], 4), round(p["frac"], 3)) for p in pl])
for key, c in sorted(seen.items(), key=lambda kv: kv[0][3]):
    t, tb, tc, a, b, cc = key
    print("tiles %6d -> retiled A %6d B %6d : row slots %8d  (seen %d x)" % (t, a, b, a * 64 + b * 32, c))
