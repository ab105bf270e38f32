"""ctypes binding of libkvmatch_hip.so (the C ABI declared in include/kvmatch.h).

There is no fallback of any kind: if the shared library is missing, or no gfx950 device is usable,
loading / kvm_open raises.  Nothing in this package imports the CPU oracle under oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (KVM_LIB in the environment: another build of the same library, e.g. a tuning variant under profiles/)
LIB_PATH = os.environ.get("KVM_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libkvmatch_hip.so")

KVM_OK = 0
KVM_ERR_INVALID_HANDLE = -1
KVM_ERR_ILLEGAL_ARGUMENT = -2
KVM_ERR_IO = -3
KVM_ERR_DEVICE = -5
KVM_ERR_NO_SERIES = -6
KVM_ERR_UNSUPPORTED = -7
KVM_ERR_INDEX_OUT_OF_BOUNDS = -8
KVM_ERR_NO_MEMORY = -9
KVM_ERR_SHARD_FAILED = -10
KVM_ABI_VERSION = 6
GROUP_ID_BYTES = 128
GROUP_RCCL_ALWAYS = 1
GROUP_PACK_CAP, GROUP_SPLIT_MODE = 1, 2
SPLIT_AUTO, SPLIT_INTERLEAVED, SPLIT_CONTIGUOUS = 0, 1, 2

STAT_WINDOWS, STAT_PASS_FILTER, STAT_PASS_LB_KIM, STAT_PASS_LB_KEOGH, STAT_PASS_LB_KEOGH2, STAT_KERNEL_NS, \
    STAT_RECHECKED, STAT_DTW_SEQUENTIAL = range(8)
OPT_EARLY_ABANDON, OPT_LB_CASCADE, OPT_ED_KERNEL, OPT_TIMING, OPT_DP_KERNEL, OPT_SORT_VIOLATION, OPT_UCR_LOOPS, \
    OPT_STRICT_BOUNDS, OPT_IN_FLIGHT = 1, 2, 3, 4, 5, 6, 7, 8, 9
OPT_ED_FORCE, OPT_SERIALIZE_ENQUEUE, OPT_ED_LAST_SHAPE = 10, 11, 12


def ed_force(split_a=0, split_b=0, a16=0, b16=0, classes=0, dev_a16_any=0, dev_no_split=0):
    """KVM_ED_FORCE of include/kvmatch.h: the value of OPT_ED_FORCE (0 = the launcher's model decides that part)."""
    return split_a | (split_b << 4) | (a16 << 8) | (b16 << 10) | (classes << 12) | (dev_a16_any << 14) | (dev_no_split << 15)


def ed_shape(value):
    """OPT_ED_LAST_SHAPE decoded: window splits, candidates per lane of classes A / B, queries in flight planned for."""
    return {"split_a": value & 15, "split_b": (value >> 4) & 15, "per_lane_a": {1: 8, 2: 16}.get((value >> 8) & 3),
            "per_lane_b": {1: 4, 2: 8}.get((value >> 10) & 3), "in_flight": (value >> 16) & 255}

# every symbol include/kvmatch.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "kvm_open", "kvm_close", "kvm_last_error", "kvm_version",
    "kvm_load_series_f64", "kvm_load_series_be64", "kvm_load_series_file", "kvm_attach_series_device", "kvm_share_series",
    "kvm_series_length", "kvm_read_time_series", "kvm_window_means", "kvm_data_envelope",
    "kvm_index_rows_build", "kvm_index_rows_count", "kvm_index_rows_keys", "kvm_index_rows_first", "kvm_index_rows_last",
    "kvm_index_rows_free", "kvm_index_rows_build_grouped", "kvm_index_rows_groups", "kvm_index_rows_group_keys",
    "kvm_index_rows_group_offsets", "kvm_index_build", "kvm_index_file_size", "kvm_index_file_bytes", "kvm_index_file_info",
    "kvm_index_file_write", "kvm_index_file_free", "kvm_phase2_time_estimate_ms", "kvm_phase2_time_estimate_dp_ms", "kvm_phase1_should_stop",
    "kvm_verify_rsm_ed", "kvm_verify_cnsm_ed", "kvm_verify_rsm_dtw", "kvm_verify_cnsm_dtw", "kvm_resolve_rho",
    "kvm_cnsm_dtw_order", "kvm_sort_and_merge_intervals", "kvm_run_jobs", "kvm_run_device_jobs",
    "kvm_result_count", "kvm_result_candidates", "kvm_result_elapsed_ms", "kvm_result_offsets",
    "kvm_result_distances", "kvm_result_stat", "kvm_result_free",
    "kvm_set_option", "kvm_get_option", "kvm_last_kernel_ns", "kvm_measure_device", "kvm_verify_device", "kvm_pack_answers_device",
    "kvm_gen_series", "kvm_gen_intervals", "kvm_write_series_file",
    "kvm_open_devices", "kvm_group_unique_id", "kvm_open_rank", "kvm_group_close", "kvm_group_last_error", "kvm_group_world",
    "kvm_group_local_devices", "kvm_group_rccl_ranks", "kvm_group_ctx", "kvm_group_configure", "kvm_group_load_series_f64",
    "kvm_group_load_series_be64", "kvm_group_load_series_file", "kvm_group_set_option", "kvm_verify_sharded",
    "kvm_run_sharded_jobs", "kvm_split_intervals", "kvm_abi_version", "kvm_group_stat",
]


class KvmJob(C.Structure):
    """struct kvm_job of include/kvmatch.h: one phase-2 call as data (kvm_run_jobs)."""
    _fields_ = [("variant", C.c_int32), ("m", C.c_int32), ("query", C.c_void_p), ("left", C.c_void_p), ("right", C.c_void_p),
                ("k", C.c_int32), ("shift", C.c_int32), ("epsilon", C.c_double), ("rho", C.c_int32), ("reserved", C.c_int32),
                ("alpha", C.c_double), ("beta", C.c_double)]


class KvmDeviceSlot(C.Structure):
    """struct kvm_device_slot of include/kvmatch.h (kvm_run_device_jobs)."""
    _fields_ = [("ctx", C.c_void_p), ("stream", C.c_void_p), ("d_off", C.c_void_p), ("d_dist", C.c_void_p), ("cap", C.c_int64),
                ("d_count", C.c_void_p), ("d_packed", C.c_void_p), ("pack_cap", C.c_int64), ("ready", C.c_void_p)]


class KvMatchDeviceError(RuntimeError):
    """HIP/runtime failure (the JNI shim maps this to java.io.IOException)."""


class KvMatchIOError(IOError):
    """I/O failure -> java.io.IOException (K/QueryEngine.java:155 `throws IOException`)."""


class KvMatchIllegalArgument(ValueError):
    """-> java.lang.IllegalArgumentException (K/operator/file/TimeSeriesFileOperator.java:55-57)."""


class KvMatchIndexOutOfBounds(IndexError):
    """-> java.lang.ArrayIndexOutOfBoundsException: where the reference's DtwUtils.lowerUpperLemire indexes below its
    arrays (K/utils/DtwUtils.java:81-90; rho >= m, or an interval whose fetched data holds <= rho points).  Raised with
    OPT_STRICT_BOUNDS = 1 (default)."""


_lib = None


def load():
    """Load the shared library (once).  Raises OSError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not found: build it with `make -C kv-match_amd/csrc` "
                      "(or __graft_entry__.build()); there is no CPU fallback")
    # When torch is used in the same process (bench.py, distributed host) its bundled HIP runtime must be the one
    # that initialises the GPU: load it first so that libkvmatch_hip.so binds to the already-loaded
    # libamdhip64 (two different HIP runtimes in one process leave the second without devices).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    sig = {
        "kvm_open": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "kvm_close": (C.c_int, [vp]),
        "kvm_last_error": (C.c_char_p, [vp]),
        "kvm_version": (C.c_char_p, []),
        "kvm_load_series_f64": (C.c_int, [vp, vp, i64]),
        "kvm_load_series_be64": (C.c_int, [vp, vp, i64]),
        "kvm_load_series_file": (C.c_int, [vp, C.c_char_p, i64]),
        "kvm_attach_series_device": (C.c_int, [vp, vp, i64]),
        "kvm_share_series": (C.c_int, [vp, vp]),
        "kvm_data_envelope": (C.c_int, [vp, i32, i32, i64, vp, vp]),
        "kvm_series_length": (i64, [vp]),
        "kvm_read_time_series": (C.c_int, [vp, i32, i32, vp]),
        "kvm_window_means": (C.c_int, [vp, i32, i32, i64, vp]),
        "kvm_index_rows_build": (C.c_int, [vp, i32, C.POINTER(vp)]),
        "kvm_index_rows_count": (i64, [vp]),
        "kvm_index_rows_keys": (pd, [vp]),
        "kvm_index_rows_first": (pi, [vp]),
        "kvm_index_rows_last": (pi, [vp]),
        "kvm_index_rows_free": (None, [vp]),
        "kvm_index_rows_build_grouped": (C.c_int, [vp, i32, C.POINTER(vp)]),
        "kvm_index_rows_groups": (i64, [vp]),
        "kvm_index_rows_group_keys": (pd, [vp]),
        "kvm_index_rows_group_offsets": (pi, [vp]),
        "kvm_index_build": (C.c_int, [vp, i32, C.POINTER(vp)]),
        "kvm_index_file_size": (i64, [vp]),
        "kvm_index_file_bytes": (C.POINTER(C.c_uint8), [vp]),
        "kvm_index_file_info": (i64, [vp, i32]),
        "kvm_index_file_write": (C.c_int, [vp, C.c_char_p]),
        "kvm_index_file_free": (None, [vp]),
        "kvm_phase2_time_estimate_ms": (f64, [i32, i64, i64, i32]),
        "kvm_phase2_time_estimate_dp_ms": (f64, [i32, i64, i64, i32, i64, i32]),
        "kvm_phase1_should_stop": (i32, [i32, i32, f64, i64, i64, i32, C.POINTER(f64)]),
        "kvm_verify_rsm_ed": (C.c_int, [vp, vp, i32, vp, vp, i32, i32, f64, C.POINTER(vp)]),
        "kvm_verify_cnsm_ed": (C.c_int, [vp, vp, i32, vp, vp, i32, i32, f64, f64, f64, C.POINTER(vp)]),
        "kvm_verify_rsm_dtw": (C.c_int, [vp, vp, i32, vp, vp, i32, i32, f64, i32, C.POINTER(vp)]),
        "kvm_verify_cnsm_dtw": (C.c_int, [vp, vp, i32, vp, vp, i32, i32, f64, i32, f64, f64, C.POINTER(vp)]),
        "kvm_resolve_rho": (i32, [f64, i32]),
        "kvm_cnsm_dtw_order": (C.c_int, [vp, i32, vp]),
        "kvm_sort_and_merge_intervals": (i64, [vp, vp, i64, vp, vp]),
        "kvm_run_jobs": (C.c_int, [C.POINTER(vp), i32, C.POINTER(KvmJob), i64, C.POINTER(vp), C.POINTER(i32)]),
        "kvm_run_device_jobs": (C.c_int, [C.POINTER(KvmDeviceSlot), i32, C.POINTER(KvmJob), i64, C.POINTER(i32), C.POINTER(i32),
                                          C.POINTER(i32), C.POINTER(i64), C.POINTER(i32)]),
        "kvm_result_count": (i64, [vp]),
        "kvm_result_candidates": (i64, [vp]),
        "kvm_result_elapsed_ms": (f64, [vp]),
        "kvm_result_offsets": (pi, [vp]),
        "kvm_result_distances": (pd, [vp]),
        "kvm_result_stat": (i64, [vp, i32]),
        "kvm_result_free": (None, [vp]),
        "kvm_set_option": (C.c_int, [vp, i32, i64]),
        "kvm_get_option": (i64, [vp, i32]),
        "kvm_last_kernel_ns": (i64, [vp]),
        "kvm_measure_device": (C.c_int, [vp, i32, C.POINTER(f64)]),
        "kvm_verify_device": (C.c_int, [vp, i32, vp, i32, vp, vp, i32, i32, f64, i32, f64, f64, vp, vp, i64, vp,
                                        C.POINTER(i64), vp]),
        "kvm_pack_answers_device": (C.c_int, [vp, vp, vp, vp, i64, vp, vp]),
        "kvm_gen_series": (C.c_int, [C.c_uint64, i64, vp]),
        "kvm_gen_intervals": (i64, [C.c_uint64, i64, i32, i64, f64, vp, i32, vp, vp, i64]),
        "kvm_write_series_file": (C.c_int, [C.c_char_p, vp, i64]),
        "kvm_open_devices": (C.c_int, [vp, i32, i32, i32, C.POINTER(vp)]),
        "kvm_group_unique_id": (C.c_int, [vp, i32]),
        "kvm_open_rank": (C.c_int, [i32, i32, i32, vp, i32, i32, i32, C.POINTER(vp)]),
        "kvm_group_close": (C.c_int, [vp]),
        "kvm_group_last_error": (C.c_char_p, [vp]),
        "kvm_group_world": (i32, [vp]),
        "kvm_group_local_devices": (i32, [vp]),
        "kvm_group_rccl_ranks": (i32, [vp]),
        "kvm_group_ctx": (vp, [vp, i32, i32]),
        "kvm_group_configure": (C.c_int, [vp, i32, i64]),
        "kvm_group_load_series_f64": (C.c_int, [vp, vp, i64]),
        "kvm_group_load_series_be64": (C.c_int, [vp, vp, i64]),
        "kvm_group_load_series_file": (C.c_int, [vp, C.c_char_p, i64]),
        "kvm_group_set_option": (C.c_int, [vp, i32, i64]),
        "kvm_verify_sharded": (C.c_int, [vp, i32, vp, i32, vp, vp, i32, i32, f64, i32, f64, f64, C.POINTER(vp)]),
        "kvm_run_sharded_jobs": (C.c_int, [vp, C.POINTER(KvmJob), i64, C.POINTER(vp), C.POINTER(i32)]),
        "kvm_split_intervals": (i64, [vp, vp, i64, i32, i32, i32, vp, vp]),
        "kvm_abi_version": (i32, []),
        "kvm_group_stat": (i64, [vp, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # (struct kvm_job / kvm_device_slot are passed as arrays: a build with another layout would be mis-strided silently)
    if lib.kvm_abi_version() != KVM_ABI_VERSION:
        raise OSError(f"{LIB_PATH}: ABI version {lib.kvm_abi_version()}, this binding is written for {KVM_ABI_VERSION}")
    _lib = lib
    return lib


class KvMatchShardFailed(RuntimeError):
    """Another rank of a device group failed on its shard of the call (KVM_ERR_SHARD_FAILED; -> java.io.IOException)."""


def check(rc, ctx=None, group=None):
    """Translate a status code into the exception the reference would have thrown.  `group`: the status comes from a
    kvm_group entry point (its text is kvm_group_last_error's)."""
    if rc == KVM_OK:
        return
    msg = load().kvm_group_last_error(None if group is True else group) if group is not None else load().kvm_last_error(ctx)
    msg = msg.decode() if msg else f"status {rc}"
    if rc == KVM_ERR_SHARD_FAILED:
        raise KvMatchShardFailed(msg)
    if rc == KVM_ERR_ILLEGAL_ARGUMENT:
        raise KvMatchIllegalArgument(msg)
    if rc == KVM_ERR_IO:
        raise KvMatchIOError(msg)
    if rc == KVM_ERR_INDEX_OUT_OF_BOUNDS:
        raise KvMatchIndexOutOfBounds(msg)
    if rc == KVM_ERR_NO_MEMORY:
        raise MemoryError(msg)  # -> java.lang.OutOfMemoryError
    raise KvMatchDeviceError(f"[{rc}] {msg}")
