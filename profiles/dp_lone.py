"""Latency of ONE banded DP (m = 4096, rho = 204) and of the whole selective call around it: a list with a single candidate,
the query's own offset.  KVM_OPT_TIMING: HIP-event time of the DP kernel alone.   dp_lone.py [m]"""
import sys, time
sys.path.insert(0, "kv-match_amd")
import numpy as np, kvmatch
from kvmatch import datagen
n = 10_000_000
m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
series = datagen.gen_series(n, seed=4242)
qoff = n // 3 + 5
q = series[qoff - 1:qoff - 1 + m].copy()
ctx = kvmatch.Phase2Context(0); ctx.load_series(series)
ctx.set_option(kvmatch.OPT_TIMING, 1); ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
L = np.array([qoff], np.int32); R = L.copy()
for name, eng, args in (("rsm_dtw", kvmatch.QueryEngineDtw(ctx), (1.0, 0.05)), ("cnsm_dtw", kvmatch.NormQueryEngineDtw(ctx), (1.0, 0.05, 1.1, 1.0))):
    ks, ts = [], []
    for i in range(12):
        t = time.perf_counter(); a = eng.verify((L, R), 1, q, *args); ts.append(time.perf_counter() - t)
        ks.append(ctx._lib.kvm_last_kernel_ns(ctx._h) * 1e-6)
    print(name, "m", m, "answers", len(a), "dp kernel ms median %.4f min %.4f" % (np.median(ks[2:]), min(ks[2:])),
          "call ms median %.4f" % (np.median(ts[2:]) * 1e3), flush=True)
