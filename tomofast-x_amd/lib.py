"""ctypes binding of libtfx.so (the C ABI of include/tfx.h).

There is no CPU fallback: if the HIP library is missing or no GPU is visible, every entry point raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("TFX_LIBTFX") or os.path.join(HERE, "libtfx.so")      # (TFX_LIBTFX: another build of the same library, for A/B measurements by tools/)

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_ip = C.POINTER(C.c_int32)
c_lp = C.POINTER(C.c_int64)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALLGATHERV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p)

TFX_E = {-1: "TFX_E_ARG", -2: "TFX_E_HIP", -3: "TFX_E_GEOMETRY", -4: "TFX_E_STATE", -5: "TFX_E_NUMERIC", -6: "TFX_E_COMM"}

# every symbol include/tfx.h declares (tests check the library exports exactly these)
SYMBOLS = [
    "tfx_create", "tfx_destroy", "tfx_last_error", "tfx_device_info", "tfx_device_count", "tfx_copy", "tfx_device_malloc", "tfx_device_free", "tfx_set_allreduce", "tfx_set_allgatherv", "tfx_set_grid",
    "tfx_comm_unique_id", "tfx_comm_init_rccl", "tfx_comm_destroy", "tfx_comm_abort", "tfx_comm_info", "tfx_comm_allreduce", "tfx_comm_allgatherv", "tfx_comm_group_begin", "tfx_comm_group_end",
    "tfx_comm_send", "tfx_comm_recv", "tfx_comm_barrier",
    "tfx_column_weight_type1", "tfx_column_weight_type2", "tfx_column_weight_type3", "tfx_prism_rows_gz", "tfx_prism_rows_mag", "tfx_prism_rows", "tfx_wavelet", "tfx_compress_row",
    "tfx_build_kernel_grav", "tfx_build_kernel_mag", "tfx_build_kernel", "tfx_select_problem",
    "tfx_matrix_upload_csr", "tfx_matrix_info", "tfx_matrix_format", "tfx_matrix_download_csr", "tfx_matrix_free", "tfx_matrix_scale_rows", "tfx_matrix_normalize_columns", "tfx_matrix_reserve",
    "tfx_cons_upload_csr", "tfx_cons_clear", "tfx_rowstore_build", "tfx_rowstore_build_ex", "tfx_rowstore_build_comp", "tfx_rowstore_counts", "tfx_rowstore_pack",
    "tfx_rowstore_free", "tfx_matrix_begin", "tfx_matrix_append_rows", "tfx_matrix_finish",
    "tfx_partition_columns", "tfx_spmv", "tfx_spmtv", "tfx_lsqr_solve", "tfx_lsqr_begin", "tfx_lsqr_iterate",
    "tfx_lsqr_end", "tfx_lsqr_set_wavelet_domain", "tfx_lsqr_set_partition", "tfx_calc_data", "tfx_timer_start", "tfx_timer_stop_ms", "tfx_profile_enable", "tfx_profile_get",
    "tfx_debug_set",
    "tfx_fastmath_eval",
]


class TfxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (TFX_E.get(code, "TFX_E_?"), code, msg))
        self.code = code


_lib = None


def load():
    """Loads libtfx.so; raises (no fallback) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(SO_PATH):
        raise ImportError("libtfx.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C tomofast-x_amd/csrc`. The MI355X path has no CPU fallback." % SO_PATH)
    lib = C.CDLL(SO_PATH)
    lib.tfx_last_error.restype = C.c_char_p
    for name in SYMBOLS:
        getattr(lib, name)
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise TfxError(rc, load().tfx_last_error().decode("utf-8", "replace"))


def ptr(a):
    """numpy array -> void pointer; int -> raw (device) address; torch tensor -> its data_ptr."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    raise TypeError("unsupported buffer type %r" % type(a))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
