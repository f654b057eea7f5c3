"""ctypes binding of ``libmi_sa.so`` (include/mi_sa.h).  There is no CPU fallback: if the HIP library
is missing or cannot be loaded this module raises, loudly."""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI_SA_LIB selects another build of the same library (e.g. the -DMI_K2_PROFILE development build)
LIB_PATH = os.environ.get("MI_SA_LIB") or os.path.join(_HERE, "libmi_sa.so")

MI_OK = 0
KIND_DENSE, KIND_CSR_RANK1, KIND_POTTS_CSR = 1, 2, 3

_lock = threading.Lock()
_lib = None


class MiSaError(RuntimeError):
    """Error reported by the native engine (negative MI_E* code + message)."""

    def __init__(self, code, message):
        super().__init__("libmi_sa error %d: %s" % (code, message))
        self.code = code
        self.message = message


def _declare(lib):
    u8p, u16p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)
    i32p, f32p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)
    u64p, ip, vp = C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.c_void_p
    pp = C.POINTER(C.c_void_p)
    sig = {
        "mi_last_error": (C.c_char_p, []),
        "mi_abi_version": (C.c_int, []),
        "mi_device_count": (C.c_int, [ip]),
        "mi_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, ip, u64p]),
        "mi_sa_problem_create_dense_f32": (C.c_int, [f32p, C.c_int, C.c_double, C.c_int, pp]),
        "mi_sa_problem_create_csr_rank1_f32": (C.c_int, [i32p, i32p, f32p, f32p, C.c_float, C.c_int,
                                                         C.c_double, C.c_int, pp]),
        "mi_sa_problem_create_potts_csr_f32": (C.c_int, [i32p, i32p, f32p, C.c_float, C.c_int,
                                                         C.c_int, C.c_double, C.c_int, pp]),
        "mi_sa_problem_destroy": (C.c_int, [vp]),
        "mi_sa_problem_info": (C.c_int, [vp, ip, ip, ip, ip]),
        "mi_sa_set_option": (C.c_int, [vp, C.c_char_p, C.c_long]),
        "mi_sa_plan_slot_order": (C.c_int, [i32p, i32p, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
        "mi_sa_problem_set_absent": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8)]),
        "mi_sa_problem_set_pair_weights": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
        "mi_sa_plan_slot_layout": (C.c_int, [i32p, i32p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "mi_sa_problem_set_energy_model_f64": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double]),
        "mi_sa_debug_pace": (C.c_int, [vp, C.POINTER(C.c_uint32), C.c_int]),
        "mi_sa_debug_stats": (C.c_int, [vp, u64p, C.c_int]),
        "mi_sa_anneal": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_int, f64p, C.c_uint64, vp, C.c_int]),
        "mi_sa_anneal_ex": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_int, f64p, C.c_uint64, vp, C.c_int,
                                      C.c_uint32, C.c_uint32]),
        "mi_sa_tempering_begin": (C.c_int, [vp, f64p, C.c_int, C.c_int, C.c_uint32, C.c_int]),
        "mi_sa_tempering_exchange": (C.c_int, [vp, C.c_uint32, C.c_uint64, f64p]),
        "mi_sa_tempering_exchange_dev": (C.c_int, [vp, C.c_uint32, C.c_uint64, vp]),
        "mi_sa_device_results": (C.c_int, [vp, pp, pp, ip]),
        "mi_sa_tempering_state": (C.c_int, [vp, i32p, u64p, u64p]),
        "mi_sa_sync": (C.c_int, [vp]),
        "mi_sa_last_kernel_ms": (C.c_int, [vp, f32p]),
        "mi_sa_last_launch_count": (C.c_int, [vp, ip]),
        "mi_sa_last_kernel_name": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "mi_sa_fetch": (C.c_int, [vp, vp, f64p, u64p]),
        "mi_sa_best": (C.c_int, [vp, ip, f64p, u64p, vp]),
        "mi_multi_gpu_anneal": (C.c_int, [pp, C.c_int, C.c_int, C.c_uint32, C.c_int, f64p, C.c_uint64, C.c_int]),
        "mi_multi_gpu_best": (C.c_int, [pp, C.c_int, ip, C.POINTER(C.c_uint32), f64p, vp]),
        "mi_multi_gpu_fetch": (C.c_int, [pp, C.c_int, vp, f64p, u64p]),
        "mi_sa_qubo_dense_f32": (C.c_int, [f32p, C.c_int, C.c_double, C.c_int, C.c_int, f64p,
                                           C.c_uint64, u8p, u8p, f64p, u64p, C.c_int]),
        "mi_energy_dense_f32": (C.c_int, [f32p, C.c_int, u8p, C.c_int, C.c_double, f64p, C.c_int]),
        "mi_energy_dense_f64": (C.c_int, [f64p, C.c_int, u8p, C.c_int, C.c_double, f64p, C.c_int]),
        "mi_energy_dense_f32_ex": (C.c_int, [f32p, C.c_int, u8p, C.c_int, C.c_double, f64p, C.c_int, C.c_int,
                                             f32p]),
        # include/mi_snn.h
        "mi_snn_build_f32": (C.c_int, [f32p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, pp]),
        "mi_snn_build_ex_f32": (C.c_int, [f32p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint32, C.c_double,
                                          C.c_int, C.c_int, pp]),
        "mi_snn_build_rounded_f32": (C.c_int, [f32p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double,
                                               C.c_int, pp]),
        "mi_snn_fetch_codes": (C.c_int, [vp, u8p]),
        "mi_snn_info": (C.c_int, [vp, ip, ip, C.POINTER(C.c_int64), ip]),
        "mi_snn_fetch": (C.c_int, [vp, i32p, C.POINTER(C.c_int64), i32p, i32p]),
        "mi_snn_kernel_ms": (C.c_int, [vp, f32p, f32p, f32p]),
        "mi_snn_destroy": (C.c_int, [vp]),
        # include/mi_metrics.h
        "mi_jaccard_cluster_stats": (C.c_int, [u64p, C.c_int, C.c_int, i32p, C.c_int, C.c_int, f64p, f64p, f64p, f64p,
                                               f64p, f32p, f32p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return sig


EXPORTS = (
    "mi_last_error", "mi_abi_version", "mi_device_count", "mi_device_info",
    "mi_sa_problem_create_dense_f32", "mi_sa_problem_create_csr_rank1_f32",
    "mi_sa_problem_create_potts_csr_f32", "mi_sa_problem_destroy", "mi_sa_problem_info",
    "mi_sa_set_option", "mi_sa_plan_slot_order", "mi_sa_plan_slot_layout", "mi_sa_problem_set_absent", "mi_sa_problem_set_pair_weights", "mi_sa_problem_set_energy_model_f64", "mi_sa_debug_pace", "mi_sa_debug_stats", "mi_sa_anneal", "mi_sa_anneal_ex", "mi_sa_tempering_begin", "mi_sa_tempering_exchange", "mi_sa_tempering_exchange_dev", "mi_sa_device_results", "mi_sa_tempering_state", "mi_sa_sync", "mi_sa_last_kernel_ms", "mi_sa_last_launch_count", "mi_sa_last_kernel_name", "mi_sa_fetch", "mi_sa_best",
    "mi_multi_gpu_anneal", "mi_multi_gpu_best", "mi_multi_gpu_fetch", "mi_sa_qubo_dense_f32", "mi_energy_dense_f32", "mi_energy_dense_f64", "mi_energy_dense_f32_ex",
    "mi_snn_build_f32", "mi_snn_build_ex_f32", "mi_snn_build_rounded_f32", "mi_snn_fetch_codes", "mi_snn_info", "mi_snn_fetch", "mi_snn_kernel_ms", "mi_snn_destroy",
    "mi_jaccard_cluster_stats",
)


def load():
    """Load (once) and return the ctypes library.  PyTorch, when importable, is imported first so
    that this process holds ONE HIP runtime: torch bundles its own ``libamdhip64.so`` with the same
    SONAME that libmi_sa.so needs, and whichever is mapped first serves both."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "MI355X engine not built: %s is missing.  Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C scrna_seq_qannealing_clustering_amd/csrc`.  There is no CPU fallback."
                % LIB_PATH)
        try:
            import torch  # noqa: F401  (maps torch's HIP runtime first; see docstring)
        except Exception:
            pass
        try:
            lib = C.CDLL(LIB_PATH, mode=getattr(os, "RTLD_NOW", 2))
        except OSError as exc:
            raise RuntimeError("cannot load %s: %s (no CPU fallback exists)" % (LIB_PATH, exc)) from exc
        _declare(lib)
        _lib = lib
        return _lib


def check(code):
    if code != MI_OK:
        msg = load().mi_last_error()
        raise MiSaError(code, msg.decode("utf-8", "replace") if msg else "")
    return code


def device_count() -> int:
    n = C.c_int(0)
    rc = load().mi_device_count(C.byref(n))
    if rc != MI_OK:
        return 0
    return int(n.value)


def device_info(device: int = 0):
    name = C.create_string_buffer(256)
    cus = C.c_int(0)
    mem = C.c_uint64(0)
    check(load().mi_device_info(device, name, 256, C.byref(cus), C.byref(mem)))
    return {"name": name.value.decode(), "compute_units": int(cus.value), "hbm_bytes": int(mem.value)}
