"""Index build step 1 on the GPU (kvm_index_rows_build) at n = 1e8: device time and whole-call time per window length,
with a parity check against the oracle on the first 2e6 points' worth of rows."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kvmatch
from kvmatch import datagen
import oracle

n = int(os.environ.get("N", 100_000_000))
s = datagen.gen_series(n, seed=1)
ctx = kvmatch.Phase2Context(0)
ctx.load_series(s)
ctx.set_option(kvmatch.OPT_TIMING, 1)
out = {"n": n, "rows": []}
for w in (25, 50, 100, 200, 400):
    import ctypes as C
    def c_call():
        h = C.c_void_p()
        t0 = time.perf_counter()
        rc = ctx._lib.kvm_index_rows_build(ctx._h, w, C.byref(h))
        dt = time.perf_counter() - t0
        assert rc == 0
        cnt = int(ctx._lib.kvm_index_rows_count(h))
        ctx._lib.kvm_index_rows_free(h)
        return dt, cnt
    def g_call():
        h = C.c_void_p()
        t0 = time.perf_counter()
        rc = ctx._lib.kvm_index_rows_build_grouped(ctx._h, w, C.byref(h))
        dt = time.perf_counter() - t0
        assert rc == 0
        ng = int(ctx._lib.kvm_index_rows_groups(h))
        ctx._lib.kvm_index_rows_free(h)
        return dt, ng
    g_call()
    gdt, ng = min(g_call() for _ in range(3))
    gdev_ms = int(ctx._lib.kvm_last_kernel_ns(ctx._h)) / 1e6
    first_call, _ = c_call()
    dt, cnt = min(c_call() for _ in range(3))
    dev_ms = int(ctx._lib.kvm_last_kernel_ns(ctx._h)) / 1e6
    out["rows"].append({"w": w, "intervals": cnt, "device_ms": dev_ms, "call_ms": dt * 1e3, "first_call_ms": first_call * 1e3,
                        "positions_per_s_call": (n - w + 1) / dt, "grouped_call_ms": gdt * 1e3, "grouped_device_ms": gdev_ms, "rows": ng,
                        "positions_per_s_device": (n - w + 1) / (dev_ms / 1e3)})
    print(out["rows"][-1], flush=True)
# CPU oracle on a bounded sample + parity on it
m = min(n, 2_000_000)
ctx.load_series(s[:m])
t0 = time.perf_counter()
wk, wf, wl = oracle.index_rows(s[:m], 50)
cpu = time.perf_counter() - t0
keys, first, last = ctx.index_rows(50)
ok = np.array_equal(keys.view(np.int64), wk.view(np.int64)) and np.array_equal(first, wf) and np.array_equal(last, wl)
out["cpu_oracle_positions_per_s"] = (m - 49) / cpu
out["parity_2e6"] = bool(ok)
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "index_rows.json"), "w"), indent=1)
