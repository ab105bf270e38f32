"""CPU tests: the C-ABI library loads without a GPU and exports every symbol include/kvmatch.h declares;
no compute entry point is called here."""
import ctypes
import os
import sys
import re

import numpy as np

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "kvmatch.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kvm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = header_symbols()
    for s in ("kvm_open", "kvm_verify_rsm_ed", "kvm_verify_cnsm_ed", "kvm_verify_rsm_dtw", "kvm_verify_cnsm_dtw",
              "kvm_load_series_be64", "kvm_read_time_series", "kvm_result_offsets", "kvm_verify_device"):
        assert s in syms


def test_library_exports_every_declared_symbol(built):
    from kvmatch import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == header_symbols()
    _lib.load()
    assert _lib.load().kvm_version().startswith(b"kvmatch-hip")


def test_resolve_rho_is_host_side(built):
    from kvmatch import _lib
    lib = _lib.load()
    assert lib.kvm_resolve_rho(0.05, 4096) == 204   # K/QueryEngineDtw.java:135-140
    assert lib.kvm_resolve_rho(1.0, 100) == 100
    assert lib.kvm_resolve_rho(12.7, 100) == 12


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "kv-match_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M) or "kvm_oracle" in txt \
                        or "libkvm_oracle" in txt:
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_open_fails_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import kvmatch
    with pytest.raises(kvmatch.KvMatchDeviceError):
        kvmatch.Phase2Context(0)


def test_phase2_time_estimate_is_the_reference_formula(built):
    """kvm_phase2_time_estimate_ms: the formula of K/QueryEngine.java:319 (affine in #intervals and in #offsets * length)
    plus a per-call constant and the latency floor of the one unit that holds an answer (linear in m for ED, quadratic for
    DTW with rho = 5 percent of m) -- orders of magnitude below the Java constants; no GPU needed."""
    import kvmatch
    f = kvmatch.phase2_time_estimate_ms
    for v in range(4):
        base = f(v, 0, 0, 1024)
        assert 0.05 < base < 1.0
        a = f(v, 1000, 0, 1024) - base
        b = f(v, 0, 100_000, 1024) - base
        assert 1e-3 < a < 1e-2 and 0 < b < 0.1                       # Java: 4 071 ms resp. 276 ms for RSM-ED
        assert abs(f(v, 2000, 200_000, 1024) - (base + 2 * a + 2 * b)) < 1e-12
        # the floor term: grows with m alone (no intervals, no offsets); linearly for ED, quadratically for DTW
        d1, d2 = f(v, 0, 0, 4096) - f(v, 0, 0, 2048), f(v, 0, 0, 8192) - f(v, 0, 0, 4096)
        assert d1 > 0 and abs(d2 / d1 - (2.0 if v < 2 else 4.0)) < 1e-9
    assert f(7, 1, 1, 1) < 0


def test_phase1_should_stop_is_the_reference_loop_test(built):
    """kvm_phase1_should_stop against a transliteration of K/QueryEngine.java:316-327 driven by the same numbers: never
    before i = 5, then at the first i whose estimated total exceeds the carried one, which is not updated by the stop."""
    import kvmatch
    rng = np.random.default_rng(5)
    for trial in range(200):
        v = int(rng.integers(0, 4))
        m = int(rng.choice([256, 1024, 4096]))
        t1 = np.cumsum(rng.uniform(0.0, 0.05, 16))
        wins = np.sort(rng.integers(1, 50_000, 16))[::-1]
        offs = np.sort(rng.integers(1, 5_000_000, 16))[::-1]
        last_ref, stop_ref = 0.0, None
        for i in range(16):
            total = t1[i] + kvmatch.phase2_time_estimate_ms(v, int(wins[i]), int(offs[i]), m)
            if i >= 5 and total > last_ref:
                stop_ref = i
                break
            last_ref = total
        last, stop = 0.0, None
        for i in range(16):
            s, new = kvmatch.phase1_should_stop(v, i, float(t1[i]), int(wins[i]), int(offs[i]), m, last)
            if s:
                assert new == last
                stop = i
                break
            last = new
        assert stop == stop_ref and last == last_ref
    with pytest.raises(ValueError):
        kvmatch.phase1_should_stop(9, 0, 0.0, 1, 1, 1, 0.0)


def test_java_adapter_natives_match_the_jni_shim():
    """Every `native` method of the shipped Java adapter (kv-match_amd/java/.../Phase2Gpu.java) has its JNI export in
    kv-match_amd/jni/kvmatch_jni.c and vice versa (neither can be compiled in this image: no JDK)."""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    java = open(os.path.join(root, "kv-match_amd", "java", "cn", "edu", "fudan", "dsm", "kvmatch", "gpu", "Phase2Gpu.java")).read()
    jni = open(os.path.join(root, "kv-match_amd", "jni", "kvmatch_jni.c")).read()
    natives = set(re.findall(r"\bnative\s+[\w\[\]<>.]+\s+(\w+)\s*\(", java))
    exports = set(re.findall(r"Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_(\w+)\s*\(", jni))
    assert natives == exports and len(natives) >= 16, (natives ^ exports)
    # and every C-ABI function the shim calls is declared in the header
    header = open(os.path.join(root, "include", "kvmatch.h")).read()
    for fn in set(re.findall(r"\b(kvm_\w+)\s*\(", jni)):
        assert re.search(r"\b" + fn + r"\s*\(", header), fn


def test_jni_shim_type_checks_and_signatures_match_the_java_natives():
    """The JNI shim compiles (syntax + types, -Wall -Wextra -Werror) against a declaration-only jni.h (tests/jni_stub/: the
    JNI types and the JNIEnv entry points the shim uses, as the JNI specification declares them), and every
    Java_..._Phase2Gpu_<name> export has exactly the C signature that the `native` declaration of <name> in
    Phase2Gpu.java maps to (JNI type mapping: int -> jint, long -> jlong, double -> jdouble, String -> jstring,
    int[] -> jintArray, double[] -> jdoubleArray, objects -> jobject; static -> jclass receiver, instance -> jobject)."""
    import subprocess
    root = ROOT
    shim = os.path.join(root, "kv-match_amd", "jni", "kvmatch_jni.c")
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "tests", "jni_stub"),
                        shim], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the shim no longer holds JNI critical regions across a GPU call
    src = re.sub(r"/\*.*?\*/", "", open(shim).read(), flags=re.S)
    assert "GetPrimitiveArrayCritical" not in src
    java = open(os.path.join(root, "kv-match_amd", "java", "cn", "edu", "fudan", "dsm", "kvmatch", "gpu", "Phase2Gpu.java")).read()
    java = re.sub(r"/\*.*?\*/|//[^\n]*", "", java, flags=re.S)
    jmap = {"int": "jint", "long": "jlong", "double": "jdouble", "boolean": "jboolean", "void": "void", "String": "jstring", "int[]": "jintArray",
            "double[]": "jdoubleArray", "byte[]": "jbyteArray", "Result": "jobject"}
    natives = {}
    for mods, ret, name, params in re.findall(r"((?:public|private|static|final|\s)*)\bnative\s+([\w\[\]]+)\s+(\w+)\s*\(([^)]*)\)", java):
        ptypes = [jmap[" ".join(p.split()[:-1]).replace(" ", "")] for p in params.split(",") if p.strip()]
        natives[name] = (jmap[ret], "jclass" if "static" in mods.split() else "jobject", ptypes)
    exports = {}
    for ret, name, params in re.findall(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_(\w+)\s*\(([^)]*)\)", src):
        ps = [re.match(r"^(.*?)\w+$", x.strip()).group(1).replace(" ", "") for x in params.split(",")]
        assert ps[0] == "JNIEnv*", (name, ps)
        exports[name] = (ret, ps[1], ps[2:])
    assert set(natives) == set(exports) and len(natives) >= 16
    for name in natives:
        assert natives[name] == exports[name], (name, natives[name], exports[name])


def test_cpp_host_mirror_compiles():
    """kv-match_amd/host/kvmatch_engines.hpp (the C++ mirror of the reference's engine classes) against include/kvmatch.h."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"),
                        os.path.join(root, "kv-match_amd", "host", "kvm_cli.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_index_file_host_code_on_cpu_under_sanitizers(built, tmp_path):
    """The library's host-side index code (kvm_index_file.cpp: step 2, IndexNode.toBytesCompact, the file layout) compiled
    with AddressSanitizer + UBSan and fed the oracle's step-1 rows instead of the GPU's (tests/harness/): the image equals
    the oracle's kvo_index_file byte for byte, on CPU."""
    import subprocess
    import numpy as np
    import oracle
    from test_transliteration import _index_series
    exe = tmp_path / "index_file_harness"
    src = [os.path.join(ROOT, "tests", "harness", "index_file_harness.cpp"),
           os.path.join(ROOT, "kv-match_amd", "csrc", "kvm_index_file.cpp")]
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=all", "-pthread", "-o", str(exe)] + src, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exe_tsan = tmp_path / "index_file_harness_tsan"
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", "-fsanitize=thread", "-pthread", "-o", str(exe_tsan)] + src,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rng = np.random.default_rng(17)
    for trial in range(26):     # (the large ones: the row sequence is cut into up to 16 ranges walked by threads and stitched)
        n = int(rng.integers(300, 40_000)) if trial < 8 else (260_000 if trial < 10 else int(rng.integers(60_000, 400_000)))
        w = int(rng.choice([25, 50, 100, 200, 400]))
        s = _index_series(rng, trial % 5, n)
        keys, first, last, gk, go = oracle.index_rows_grouped(s, w)
        rows = tmp_path / "rows.bin"
        with open(rows, "wb") as f:
            f.write(np.array([len(keys), len(gk)], dtype=np.int64).tobytes())
            for arr, dt in ((keys, np.float64), (first, np.int32), (last, np.int32), (gk, np.float64), (go, np.int32)):
                f.write(np.ascontiguousarray(arr, dtype=dt).tobytes())
        out = tmp_path / "image.out"
        want, info = oracle.index_file(s, w)
        # as the library runs it, and with the thresholds of the pipelined walk (kvm_index_file.cpp, run_team: two or four threads, each
        # merging into the union its predecessor is still writing) lowered so that these small lists go through it -- under
        # ASan / UBSan, and every fourth trial under ThreadSanitizer as well
        runs = [(exe, []), (exe, ["8", "1", "2"]), (exe, ["8", "1", "4"])] + ([(exe_tsan, ["8", "1", "2" if trial % 8 == 1 else "4"])] if trial % 4 == 1 else [])
        for binary, extra in runs:
            out.unlink(missing_ok=True)
            r = subprocess.run([str(binary), str(rows), str(out)] + extra, capture_output=True, text=True,
                               env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1"))
            assert r.returncode == 0, (trial, extra, r.stderr[-2000:])
            assert out.read_bytes() == want, (trial, n, w, extra)
            assert [int(x) for x in r.stdout.split()] == info
    # no rows at all -> the IllegalArgument path (the Java code dies on rawStatisticInfo.get(0))
    with open(rows, "wb") as f:
        f.write(np.array([0, 0], dtype=np.int64).tobytes() + np.zeros(1, dtype=np.int32).tobytes())
    assert subprocess.run([str(exe), str(rows), str(out)], capture_output=True).returncode == 3


def test_timsort_restatement_is_memory_safe_under_any_comparator(tmp_path):
    """kvm_jsort.h under AddressSanitizer + UBSan with comparators that obey nothing (tests/harness/fuzz_jsort.cpp): in a
    JVM a broken comparator ends in an exception at worst; the C++ restatement must not touch memory out of bounds either."""
    import subprocess
    exe = tmp_path / "fuzz_jsort"
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=all", "-o", str(exe), os.path.join(ROOT, "tests", "harness", "fuzz_jsort.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    ok, thrown = int(r.stdout.split()[1]), int(r.stdout.split()[3])
    assert ok > 500 and thrown > 500


def test_ed_launch_model_on_the_cpu(tmp_path):
    """The ED pruner launcher's model of a launch (kvm_ed_launch_model.h: window splits, tile classes) is host-only code on
    every query's critical path: compiled alone with g++ -Wall -Wextra -Werror under ASan + UBSan and checked against the
    choices the sweeps on the MI355X measured as best (profiles/r02_ed_split_sweep_*.log), for monotonicity, for
    degenerate inputs (no units, millions of units, no wave slots) and for its cost per decision."""
    import subprocess
    exe = tmp_path / "launch_model_harness"
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "kv-match_amd", "csrc"), "-o", str(exe),
                        os.path.join(ROOT, "tests", "harness", "launch_model_harness.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "FAIL" not in r.stdout and r.stdout.count("ok  ") >= 16, r.stdout + r.stderr[-2000:]


def test_stream_host_layer_under_thread_and_address_sanitizers(tmp_path):
    """kvm_stream.cpp (kvm_run_jobs, kvm_run_device_jobs, the persistent worker pool) is host-only code that several threads
    run through at once: compiled as it is -- with stand-ins for the library entry points its loops call
    (tests/harness/stream_harness.cpp) -- under ThreadSanitizer and under ASan + UBSan.  Checked: a context is never
    entered twice at once, results and statuses per job, the first failing job's text ends up in the first context,
    several callers share the pool, the produced / consumed hand-shake of the device-resident form incl. a failing slot, and
    a fork()ed child gets a pool of its own."""
    import subprocess
    src = [os.path.join(ROOT, "tests", "harness", "stream_harness.cpp"), os.path.join(ROOT, "kv-match_amd", "csrc", "kvm_stream.cpp")]
    for san, arg, checks in (("thread", "tsan", 9), ("address,undefined", "fork", 10)):
        exe = tmp_path / ("stream_harness_" + arg)
        r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsanitize=" + san,
                            "-fno-sanitize-recover=all", "-pthread", "-o", str(exe)] + src, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        r = subprocess.run([str(exe), arg], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "FAIL" not in r.stdout and r.stdout.count("ok  ") == checks, r.stdout + r.stderr[-3000:]
        assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr, r.stderr[-3000:]


def test_no_scalar_select_on_a_stale_scc_in_the_device_code(tmp_path):
    """A miscompile met in round 5 (hipcc of ROCm 7.2, kvm_ed_dot.hip): a wave-uniform condition kept in a VGPR, `v_cmp`
    into VCC, then `s_cselect_b32` -- which reads SCC -- with the `s_and_b64 vcc, vcc, exec` between them dropped: the select
    took the SCC of the exec-mask restore in front of it and the pruner's units started every 8 tiles while covering 16
    (answers lost and duplicated; caught by the parity suite).  Every HIP source is compiled to gfx950 assembly and
    scanned for an SCC reader fed by an exec-mask operation (profiles/tools/scc_lint.py)."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(ROOT, "kv-match_amd", "csrc")
    srcs = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    # (the flags of the shipped build, from its Makefile -- not a hand-kept copy that can drift from it)
    mk = subprocess.run(["make", "-pn", "-C", csrc, "libkvmatch_hip.so"], capture_output=True, text=True).stdout
    hipflags = next(l.split("=", 1)[1].split() for l in mk.splitlines() if l.startswith("HIPFLAGS :="))   # (printed expanded)
    flags = [f for f in hipflags if not f.startswith("-W")] + ["-S", "--cuda-device-only"]
    assert "--offload-arch=gfx950" in flags and "-ffp-contract=off" in flags and "-O3" in flags, flags

    def asm(f):
        out = str(tmp_path / (f[:-4] + ".s"))
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-o", out, os.path.join(csrc, f)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return out
    with ThreadPoolExecutor(max_workers=6) as ex:
        outs = list(ex.map(asm, srcs))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "tools", "scc_lint.py")] + outs, capture_output=True, text=True)
    assert r.returncode == 0 and "0 stale-SCC suspects" in r.stdout, r.stdout[-3000:]
    # the linter does see the pattern
    bad = tmp_path / "bad.s"
    bad.write_text("f:\n\ts_or_b64 exec, exec, s[0:1]\n\tv_cmp_eq_u32_e32 vcc, 0, v128\n\ts_cselect_b32 s0, 3, 4\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "tools", "scc_lint.py"), str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "1 stale-SCC suspects" in r.stdout, r.stdout


def test_sort_and_merge_intervals_is_the_reference_step_between_the_phases(built):
    """kvm_sort_and_merge_intervals (host only) == QueryEngine.sortAndMergeIntervals (K/QueryEngine.java:664-693)."""
    import numpy as np
    import kvmatch
    import java_transliteration as J
    rng = np.random.default_rng(8)
    for trial in range(300):
        k = int(rng.integers(0, 60))
        left = rng.integers(1, 400, k)
        right = left + rng.integers(0, 30, k)
        want = J.sort_and_merge_intervals([(int(a), int(b), 1.0) for a, b in zip(left, right)])
        gl, gr = kvmatch.sort_and_merge_intervals((left, right))
        assert list(zip(gl.tolist(), gr.tolist())) == [(a, b) for a, b, _ in want], trial
        if k > 1:     # sorted, disjoint, non-adjacent: the form phase 2 takes
            assert np.all(gl[1:] > gr[:-1] + 1)
    gl, gr = kvmatch.sort_and_merge_intervals((np.array([5, 1, 3]), np.array([6, 2, 3])))
    assert gl.tolist() == [1, 5] and gr.tolist() == [3, 6]        # [1,2] and [3,3] are adjacent and joined; [5,6] stays apart


def test_interval_argument_forms_cannot_be_misread(built):
    """kvmatch.engine._as_intervals: pairs, a (k, 2) array and a 2-tuple of numpy arrays are the three accepted forms; a tuple
    of two plain LISTS of k != 2 entries -- which rounds 1-2 read as (left[], right[]) -- is refused instead of being re-read
    as k pairs in the wrong order; a tuple of two pairs is two intervals (documented in INTEGRATION.md section 6)."""
    import kvmatch
    from kvmatch.engine import _as_intervals
    l, r = _as_intervals([(1, 5), (10, 20), (30, 40)])
    assert l.tolist() == [1, 10, 30] and r.tolist() == [5, 20, 40] and l.dtype == np.int32
    l, r = _as_intervals(np.array([[1, 5], [10, 20]]))
    assert l.tolist() == [1, 10] and r.tolist() == [5, 20]
    l, r = _as_intervals((np.array([1, 10, 30]), np.array([5, 20, 40])))
    assert l.tolist() == [1, 10, 30] and r.tolist() == [5, 20, 40]
    l, r = _as_intervals(((1, 5), (10, 20)))
    assert l.tolist() == [1, 10] and r.tolist() == [5, 20]
    l, r = _as_intervals([])
    assert l.size == 0 and r.size == 0
    with pytest.raises(kvmatch.KvMatchIllegalArgument):
        _as_intervals(([1, 10, 30], [5, 20, 40]))
    with pytest.raises(kvmatch.KvMatchIllegalArgument):
        _as_intervals([1, 5, 10, 20])


def test_no_cpp_exception_crosses_the_c_abi(built, tmp_path):
    """The extern "C" entry points run their bodies behind kvm_guard (csrc/kvm_guard.h): a std::bad_alloc inside the library
    comes back as KVM_ERR_NO_MEMORY (-> MemoryError here, OutOfMemoryError in the JNI shim) instead of unwinding into the
    caller's frame -- which would end a JVM, and ends a ctypes caller with SIGABRT.  Forced in a child process whose address
    space is capped just above what it already uses, on a host-only entry point (no GPU): sortAndMergeIntervals of 2e7
    intervals needs ~0.5 GB of vectors."""
    import subprocess
    import sys
    code = r"""
import resource, sys
sys.path.insert(0, %r)
import numpy as np
from kvmatch import _lib
lib = _lib.load()
k = 20_000_000
left = np.arange(1, 3 * k, 3, dtype=np.int32)[:k].copy(); right = left + 1
ol = np.empty_like(left); orr = np.empty_like(right)
assert lib.kvm_sort_and_merge_intervals(left.ctypes.data, right.ctypes.data, 1000, ol.ctypes.data, orr.ctypes.data) == 1000
with open("/proc/self/statm") as f: vm = int(f.read().split()[0]) * resource.getpagesize()
resource.setrlimit(resource.RLIMIT_AS, (vm + (96 << 20), vm + (96 << 20)))
rc = lib.kvm_sort_and_merge_intervals(left.ctypes.data, right.ctypes.data, k, ol.ctypes.data, orr.ctypes.data)
print("rc", rc)
try:
    _lib.check(int(rc), None)
except MemoryError as e:
    print("MemoryError", e)
""" % os.path.join(ROOT, "kv-match_amd")
    cp = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, (cp.returncode, cp.stdout[-500:], cp.stderr[-2000:])
    assert "rc -9" in cp.stdout and "MemoryError" in cp.stdout, cp.stdout


def test_integration_listing_is_the_shipped_adapter():
    """INTEGRATION.md section 2 shows the Java adapter a maintainer adds: the listing is the shipped source file from its
    `package` line on, verbatim -- so the document cannot lag the natives the JNI shim implements."""
    java = open(os.path.join(ROOT, "kv-match_amd", "java", "cn", "edu", "fudan", "dsm", "kvmatch", "gpu", "Phase2Gpu.java")).read()
    body = java[java.index("package cn.edu"):].rstrip() + "\n"
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    a = doc.index("```java\npackage cn.edu.fudan.dsm.kvmatch.gpu;") + len("```java\n")
    assert doc[a: doc.index("```\n", a)] == body


def test_run_jobs_rejects_bad_arguments_without_a_gpu(built):
    """kvm_run_jobs' argument checks run before any GPU call: null context arrays / handles, negative counts, missing result
    arrays come back as status codes (no device is needed, none is touched); the Python Job marshals what verify() would
    pass (shift = (lastSegment - 1) * 25, rho resolved like the REPL's prompt)."""
    import ctypes as C
    import kvmatch
    from kvmatch import _lib
    lib = _lib.load()
    jobs = (_lib.KvmJob * 1)()
    res = (C.c_void_p * 1)()
    st = (C.c_int32 * 1)()
    null_ctx = (C.c_void_p * 1)(None)
    assert lib.kvm_run_jobs(None, 1, jobs, 1, res, st) == _lib.KVM_ERR_ILLEGAL_ARGUMENT
    assert lib.kvm_run_jobs(null_ctx, 0, jobs, 1, res, st) == _lib.KVM_ERR_ILLEGAL_ARGUMENT
    assert lib.kvm_run_jobs(null_ctx, 1, jobs, -1, res, st) == _lib.KVM_ERR_ILLEGAL_ARGUMENT
    assert lib.kvm_run_jobs(null_ctx, 1, None, 1, res, st) == _lib.KVM_ERR_ILLEGAL_ARGUMENT
    assert lib.kvm_run_jobs(null_ctx, 1, jobs, 1, None, st) == _lib.KVM_ERR_ILLEGAL_ARGUMENT
    assert lib.kvm_run_jobs(null_ctx, 1, jobs, 1, res, st) == _lib.KVM_ERR_INVALID_HANDLE
    assert C.sizeof(_lib.KvmJob) == 72        # struct kvm_job of include/kvmatch.h (two int32, three pointers, ... on LP64)
    j = kvmatch.Job(3, (np.array([5, 40], dtype=np.int32), np.array([9, 44], dtype=np.int32)), 3, np.arange(200.0), 2.5, rho=0.05,
                    alpha=1.5, beta=2.0)
    assert (j.variant, j.shift, j.rho, j.epsilon, j.alpha, j.beta, j.q.size, j.left.tolist()) == (3, 50, 10, 2.5, 1.5, 2.0, 200, [5, 40])
    assert kvmatch.Job(0, [(1, 2)], 1, [1.0] * 30, 1.0).rho == 0
