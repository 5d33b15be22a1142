"""ctypes binding of libmi355dr.so (the C ABI in include/mi355dr.h).

There is deliberately NO CPU fallback here: if the shared library is missing or no MI355X is
visible, every entry point raises.  The CPU oracle lives in oracle/ and is test infrastructure.
"""

from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = _PKG_DIR / "libmi355dr.so"

_lib: ctypes.CDLL | None = None

# every symbol include/mi355dr.h declares (checked by tests/test_abi_symbols.py)
ABI_SYMBOLS = [
    "mi355dr_create", "mi355dr_destroy", "mi355dr_last_error", "mi355dr_version", "mi355dr_reserve",
    "mi355dr_add_rows", "mi355dr_add_rows_device", "mi355dr_size", "mi355dr_dim", "mi355dr_get_rows",
    "mi355dr_search", "mi355dr_search_device", "mi355dr_search_device_async", "mi355dr_search_wait", "mi355dr_add_multivec", "mi355dr_size_multivec",
    "mi355dr_search_maxsim", "mi355dr_search_maxsim_device", "mi355dr_maxsim_subset", "mi355dr_maxsim_subset_ex", "mi355dr_add_multivec_device", "mi355dr_gqr_refine", "mi355dr_gqr_refine_maxsim",
    "mi355dr_gqr_refine_scores", "mi355dr_merge_topk_device", "mi355dr_pack_topk_device",
    "mi355dr_merge_topk_packed_device", "mi355dr_comm_unique_id", "mi355dr_comm_init", "mi355dr_comm_world", "mi355dr_comm_count", "mi355dr_comm_init_custom",
    "mi355dr_search_sharded_device", "mi355dr_set_option", "mi355dr_get_stat",
    "mi355dr_reset_stats", "mi355dr_timer_start", "mi355dr_timer_stop", "mi355dr_synchronize",
    "mi355dr_dev_alloc", "mi355dr_dev_free", "mi355dr_dev_upload", "mi355dr_dev_download",
    "mi355dr_diag_mfma_stream",
    "mi355dr_debug_screen_dense", "mi355dr_debug_screen_bound", "mi355dr_debug_i8_state", "mi355dr_debug_rescore",
]


# mi355dr_allgather_fn (include/mi355dr.h): (send_dev, recv_dev, bytes_per_rank, stream, user) -> 0 on success
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p)


class NativeError(RuntimeError):
    """A libmi355dr call failed (code + the library's last_error text)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libmi355dr error {code}: {message}")
        self.code = code


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 (same
    sonames as /opt/rocm's).  If libmi355dr.so pulled in the system copies first and torch is imported later, torch
    comes up with "No HIP GPUs are available"; the other order works (torch's copies are found by soname).  So when
    torch is installed, its copies are loaded here -- without importing torch -- before libmi355dr.so."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = Path(list(spec.submodule_search_locations)[0]) / "lib"
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        f = libdir / name
        if f.exists():
            try:
                ctypes.CDLL(str(f), mode=getattr(os, "RTLD_NOW", 2) | getattr(os, "RTLD_GLOBAL", 0x100))
            except OSError:
                return


def load() -> ctypes.CDLL:
    """Load libmi355dr.so and declare argument types.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the search path."
        )
    _preload_torch_hip_runtime()
    L = ctypes.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
    vp, c_int, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    f32p, f64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
    i64p, i32p = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)
    L.mi355dr_create.restype = c_int
    L.mi355dr_create.argtypes = [ctypes.POINTER(vp), c_int, c_int, c_int]
    L.mi355dr_destroy.restype = None
    L.mi355dr_destroy.argtypes = [vp]
    L.mi355dr_last_error.restype = ctypes.c_char_p
    L.mi355dr_last_error.argtypes = [vp]
    L.mi355dr_version.restype = c_int
    L.mi355dr_reserve.restype = c_int
    L.mi355dr_reserve.argtypes = [vp, i64]
    L.mi355dr_add_rows.restype = c_int
    L.mi355dr_add_rows.argtypes = [vp, f32p, i64]
    L.mi355dr_add_rows_device.restype = c_int
    L.mi355dr_add_rows_device.argtypes = [vp, vp, i64]
    L.mi355dr_size.restype = i64
    L.mi355dr_size.argtypes = [vp]
    L.mi355dr_dim.restype = c_int
    L.mi355dr_dim.argtypes = [vp]
    L.mi355dr_get_rows.restype = c_int
    L.mi355dr_get_rows.argtypes = [vp, i64, i64, f32p]
    L.mi355dr_search.restype = c_int
    L.mi355dr_search.argtypes = [vp, f32p, c_int, c_int, f64p, i64p]
    L.mi355dr_search_device.restype = c_int
    L.mi355dr_search_device.argtypes = [vp, vp, c_int, c_int, vp, vp, vp]
    L.mi355dr_search_device_async.restype = c_int
    L.mi355dr_search_device_async.argtypes = [vp, vp, c_int, c_int, vp, vp, vp, i64p]
    L.mi355dr_search_wait.restype = c_int
    L.mi355dr_search_wait.argtypes = [vp, i64]
    L.mi355dr_add_multivec.restype = c_int
    L.mi355dr_add_multivec.argtypes = [vp, f32p, i64p, i64]
    L.mi355dr_size_multivec.restype = i64
    L.mi355dr_size_multivec.argtypes = [vp]
    L.mi355dr_search_maxsim.restype = c_int
    L.mi355dr_search_maxsim.argtypes = [vp, f32p, i32p, c_int, c_int, f32p, i64p]
    L.mi355dr_search_maxsim_device.restype = c_int
    L.mi355dr_search_maxsim_device.argtypes = [vp, vp, i32p, c_int, c_int, vp, vp, vp]
    L.mi355dr_maxsim_subset.restype = c_int
    L.mi355dr_maxsim_subset.argtypes = [vp, f32p, i32p, c_int, i64p, c_int, f32p]
    L.mi355dr_maxsim_subset_ex.restype = c_int
    L.mi355dr_maxsim_subset_ex.argtypes = [vp, f32p, i32p, c_int, i64p, c_int, c_int, f32p]
    L.mi355dr_add_multivec_device.restype = c_int
    L.mi355dr_add_multivec_device.argtypes = [vp, vp, i64p, i64]
    c_double = ctypes.c_double
    L.mi355dr_gqr_refine.restype = c_int
    L.mi355dr_gqr_refine.argtypes = [vp, f64p, c_int, i64p, c_int, f64p, c_int, c_double, c_double, c_double, f64p]
    L.mi355dr_gqr_refine_maxsim.restype = c_int
    L.mi355dr_gqr_refine_maxsim.argtypes = [vp, f64p, i32p, c_int, i64p, c_int, f64p, c_int, c_double, c_double, c_double,
                                            f64p]
    L.mi355dr_gqr_refine_scores.restype = c_int
    L.mi355dr_gqr_refine_scores.argtypes = [vp, f64p, i32p, c_int, c_int, f64p, c_int, c_double, c_double, c_double, f64p]
    L.mi355dr_merge_topk_device.restype = c_int
    L.mi355dr_merge_topk_device.argtypes = [vp, vp, vp, c_int, c_int, c_int, vp, vp, vp]
    L.mi355dr_pack_topk_device.restype = c_int
    L.mi355dr_pack_topk_device.argtypes = [vp, vp, vp, c_int, c_int, vp, vp]
    L.mi355dr_merge_topk_packed_device.restype = c_int
    L.mi355dr_merge_topk_packed_device.argtypes = [vp, vp, c_int, c_int, c_int, vp, vp, vp]
    L.mi355dr_comm_unique_id.restype = c_int
    L.mi355dr_comm_unique_id.argtypes = [vp, ctypes.c_size_t]
    L.mi355dr_comm_init.restype = c_int
    L.mi355dr_comm_init.argtypes = [vp, c_int, c_int, vp, ctypes.c_size_t]
    L.mi355dr_comm_init_custom.restype = c_int
    L.mi355dr_comm_init_custom.argtypes = [vp, c_int, c_int, ALLGATHER_FN, vp]
    L.mi355dr_comm_world.restype = c_int
    L.mi355dr_comm_world.argtypes = [vp]
    L.mi355dr_comm_count.restype = c_int
    L.mi355dr_comm_count.argtypes = [vp, ctypes.POINTER(c_int)]
    L.mi355dr_search_sharded_device.restype = c_int
    L.mi355dr_search_sharded_device.argtypes = [vp, vp, c_int, c_int, vp, vp, vp]
    L.mi355dr_set_option.restype = c_int
    L.mi355dr_set_option.argtypes = [vp, ctypes.c_char_p, i64]
    L.mi355dr_get_stat.restype = c_int
    L.mi355dr_get_stat.argtypes = [vp, ctypes.c_char_p, i64p]
    L.mi355dr_reset_stats.restype = c_int
    L.mi355dr_reset_stats.argtypes = [vp]
    L.mi355dr_timer_start.restype = c_int
    L.mi355dr_timer_start.argtypes = [vp]
    L.mi355dr_timer_stop.restype = c_int
    L.mi355dr_timer_stop.argtypes = [vp, f64p]
    L.mi355dr_synchronize.restype = c_int
    L.mi355dr_synchronize.argtypes = [vp]
    L.mi355dr_dev_alloc.restype = c_int
    L.mi355dr_dev_alloc.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    L.mi355dr_dev_free.restype = c_int
    L.mi355dr_dev_free.argtypes = [vp, vp]
    L.mi355dr_dev_upload.restype = c_int
    L.mi355dr_dev_upload.argtypes = [vp, vp, vp, ctypes.c_size_t]
    L.mi355dr_dev_download.restype = c_int
    L.mi355dr_dev_download.argtypes = [vp, vp, vp, ctypes.c_size_t]
    L.mi355dr_debug_screen_dense.restype = c_int
    L.mi355dr_debug_screen_dense.argtypes = [vp, f32p, c_int, i64, i64, f32p]
    L.mi355dr_debug_screen_bound.restype = c_int
    L.mi355dr_debug_screen_bound.argtypes = [vp, f32p, c_int, f32p]
    L.mi355dr_debug_i8_state.restype = c_int
    L.mi355dr_debug_i8_state.argtypes = [vp, f32p, c_int, f32p, f32p, i64, i64, f32p, f32p]
    L.mi355dr_debug_rescore.restype = c_int
    L.mi355dr_debug_rescore.argtypes = [vp, f32p, c_int, i32p, i64p, i64, f32p, f64p]
    L.mi355dr_diag_mfma_stream.restype = c_int
    L.mi355dr_diag_mfma_stream.argtypes = [c_int, c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    _lib = L
    return L


def diag_mfma_stream(device: int, fmt: str, seconds: float) -> float:
    """Bare MFMA stream rate in TOP/s (`fmt`: "i8" | "bf16") measured for `seconds` on `device` (mi355dr_diag_mfma_stream)."""
    out = ctypes.c_double(0.0)
    rc = load().mi355dr_diag_mfma_stream(int(device), {"i8": 0, "bf16": 1}[fmt], float(seconds), ctypes.byref(out))
    if rc != 0:
        raise NativeError(rc, "mi355dr_diag_mfma_stream failed")
    return float(out.value)


def check(handle, rc: int) -> None:
    if rc != 0:
        msg = load().mi355dr_last_error(handle)
        raise NativeError(rc, msg.decode("utf-8", "replace") if msg else "")


def f32c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def ptr(a: np.ndarray, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))
