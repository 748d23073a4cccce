"""ctypes binding of libvidc.so (include/vidc.h).  Thin: no compute happens in Python.

The library is hand-written HIP for gfx950; there is no CPU fallback.  If libvidc.so is missing it
is built with hipcc (build.py); if it cannot be loaded, or no HIP device exists, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvidc.so")

VIDC_PREC_REFERENCE = -1
VIDC_PREC_EXACT = -2
VIDC_ROC_WANT_PERM = 1
VIDC_EF_WANT_PERM = 1
VIDC_ROC_MAX_LIST = 262144

_lib = None

_vp = C.c_void_p
_u64 = C.c_uint64
_u32 = C.c_uint32
_P = C.POINTER


class VidcError(RuntimeError):
    """Raised for every non-zero vidc_status (mirrors FaissException -> RuntimeError, custom_invlists.swig:38-57)."""


def _declare(lib):
    def f(name, restype, *argtypes):
        fn = getattr(lib, name)  # AttributeError here = libvidc.so is stale / incomplete: rebuild
        fn.restype = restype
        fn.argtypes = list(argtypes)

    f("vidc_last_error", C.c_char_p)
    f("vidc_version", C.c_int)
    f("vidc_ctx_create", C.c_int, C.c_int, _P(_vp))
    f("vidc_ctx_destroy", None, _vp)
    f("vidc_ctx_set_stream", C.c_int, _vp, _vp)
    f("vidc_ctx_reset_stream", C.c_int, _vp)
    f("vidc_ctx_synchronize", C.c_int, _vp)
    f("vidc_ctx_trim", C.c_int, _vp, C.POINTER(C.c_uint64))
    f("vidc_ctx_debug_pool_poison", C.c_int, _vp, C.c_int)
    f("vidc_ctx_class_streams", C.c_int, _vp)
    f("vidc_dev_alloc", C.c_int, _vp, C.c_size_t, _P(_vp))
    f("vidc_dev_free", C.c_int, _vp, _vp)
    f("vidc_copy_h2d", C.c_int, _vp, _vp, _vp, C.c_size_t)
    f("vidc_copy_d2h", C.c_int, _vp, _vp, _vp, C.c_size_t)
    f("vidc_ctx_d2h_bytes", _u64, _vp)
    f("vidc_ctx_last_kernel_ms", C.c_double, _vp)
    f("vidc_ctx_phase_ms", C.c_double, _vp, C.c_int)
    f("vidc_ctx_chain_info", C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp)
    # ROC
    f("vidc_roc_encode", C.c_int, _vp, _u64, _vp, _vp, C.c_int, _u32, _P(_vp))
    f("vidc_roc_encode_rows", C.c_int, _vp, _u64, _u32, _vp, C.c_int, _u32, _P(_vp))
    f("vidc_roc_destroy", None, _vp)
    f("vidc_roc_nlist", _u64, _vp)
    f("vidc_roc_ntotal", _u64, _vp)
    f("vidc_roc_compressed_bytes", _u64, _vp)
    f("vidc_roc_total_words", _u64, _vp)
    f("vidc_roc_list_info", C.c_int, _vp, _vp, _vp, _vp, _vp, _vp)
    f("vidc_roc_export_words", C.c_int, _vp, _vp, _u64, _vp, C.c_size_t)
    f("vidc_roc_export_all_words", C.c_int, _vp, _vp, _vp, C.c_size_t)
    f("vidc_roc_perm", C.c_int, _vp, _vp, _vp)
    f("vidc_roc_perm_dev", _vp, _vp)
    f("vidc_roc_import", C.c_int, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _P(_vp))
    f("vidc_roc_decode_all", C.c_int, _vp, _vp, _vp)
    f("vidc_roc_decode_lists", C.c_int, _vp, _vp, _u64, _vp, _vp, _vp)
    f("vidc_roc_decode_gather", C.c_int, _vp, _vp, _u64, _vp, _u64, _vp, _vp, _vp)
    f("vidc_roc_decode_rows", C.c_int, _vp, _vp, _u64, _vp, _u32, _vp, _vp)
    f("vidc_roc_last_decode_nonclean", _u64, _vp)
    # packed bits
    f("vidc_packed_bits_for", C.c_int, _u64)
    f("vidc_packed_encode", C.c_int, _vp, _u64, _vp, _vp, C.c_int, _P(_vp))
    f("vidc_packed_destroy", None, _vp)
    f("vidc_packed_compressed_bytes", _u64, _vp)
    f("vidc_packed_bits", C.c_int, _vp)
    f("vidc_packed_decode_all", C.c_int, _vp, _vp, _vp)
    f("vidc_packed_decode_lists", C.c_int, _vp, _vp, _u64, _vp, _vp, _vp)
    f("vidc_packed_decode_gather", C.c_int, _vp, _vp, _u64, _vp, _u64, _vp, _vp, _vp)
    f("vidc_packed_get", C.c_int, _vp, _vp, _u64, _vp, _vp, _vp)
    f("vidc_packed_export", C.c_int, _vp, _vp, _u64, _vp, C.c_size_t)
    f("vidc_packed_total_words", _u64, _vp)
    f("vidc_packed_export_all", C.c_int, _vp, _vp, _vp, C.c_size_t)
    f("vidc_packed_import", C.c_int, _vp, _u64, _vp, C.c_int, _vp, _u64, _P(_vp))
    # Elias-Fano
    f("vidc_ef_encode", C.c_int, _vp, _u64, _vp, _vp, _u32, _P(_vp))
    f("vidc_ef_destroy", None, _vp)
    f("vidc_ef_compressed_bytes", _u64, _vp)
    f("vidc_ef_list_info", C.c_int, _vp, _vp, _vp, _vp)
    f("vidc_ef_decode_all", C.c_int, _vp, _vp, _vp)
    f("vidc_ef_get", C.c_int, _vp, _vp, _u64, _vp, _vp, _vp)
    f("vidc_ef_perm", C.c_int, _vp, _vp, _vp)
    f("vidc_ef_export", C.c_int, _vp, _vp, _u64, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp)
    f("vidc_ef_stream_words", C.c_int, _vp, _vp, _vp)
    f("vidc_ef_export_all", C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t)
    f("vidc_ef_import", C.c_int, _vp, _u64, _vp, _vp, _vp, _vp, _u64, _vp, _u64, _P(_vp))
    f("vidc_ef_encode_rows", C.c_int, _vp, _u64, _u32, _vp, _P(_vp))
    f("vidc_ef_decode_rows", C.c_int, _vp, _vp, _u64, _vp, _u32, _vp, _vp)
    f("vidc_ef_decode_lists", C.c_int, _vp, _vp, _u64, _vp, _vp, _vp)
    f("vidc_ef_decode_gather", C.c_int, _vp, _vp, _u64, _vp, _u64, _vp, _vp, _vp)
    # compact graph rows
    f("vidc_compact_rows_encode", C.c_int, _vp, _u64, _u32, _vp, _P(_vp))
    f("vidc_compact_destroy", None, _vp)
    f("vidc_compact_bits", _u32, _vp)
    f("vidc_compact_stride", _u32, _vp)
    f("vidc_compact_size_in_bytes", _u64, _vp)
    f("vidc_compact_rows_decode", C.c_int, _vp, _vp, _u64, _vp, _vp, _vp)
    f("vidc_compact_export_row", C.c_int, _vp, _vp, _u64, _vp, C.c_size_t)
    # wavelet tree
    f("vidc_wt_build", C.c_int, _vp, _u64, _vp, _vp, C.c_int, _P(_vp))
    f("vidc_wt_destroy", None, _vp)
    f("vidc_wt_size_in_bytes", _u64, _vp)
    f("vidc_wt_levels", _u32, _vp)
    f("vidc_wt_select", C.c_int, _vp, _vp, _u64, _vp, _vp, _vp)
    f("vidc_wt_decode_all", C.c_int, _vp, _vp, _vp)
    f("vidc_wt_decode_lists", C.c_int, _vp, _vp, _u64, _vp, _vp, _vp)
    f("vidc_wt_decode_gather", C.c_int, _vp, _vp, _u64, _vp, _u64, _vp, _vp, _vp)


#: every symbol include/vidc.h declares (checked by the CPU test-suite against the built library)
EXPORTED_SYMBOLS = [
    "vidc_last_error", "vidc_version", "vidc_ctx_create", "vidc_ctx_destroy", "vidc_ctx_set_stream", "vidc_ctx_reset_stream",
    "vidc_ctx_synchronize", "vidc_ctx_trim", "vidc_ctx_debug_pool_poison", "vidc_ctx_class_streams", "vidc_dev_alloc", "vidc_dev_free", "vidc_copy_h2d", "vidc_copy_d2h", "vidc_ctx_d2h_bytes",
    "vidc_ctx_last_kernel_ms", "vidc_ctx_phase_ms", "vidc_ctx_chain_info",
    "vidc_roc_encode", "vidc_roc_encode_rows", "vidc_roc_destroy", "vidc_roc_nlist", "vidc_roc_ntotal",
    "vidc_roc_compressed_bytes", "vidc_roc_total_words", "vidc_roc_list_info", "vidc_roc_export_words", "vidc_roc_export_all_words",
    "vidc_roc_perm", "vidc_roc_perm_dev", "vidc_roc_import", "vidc_roc_decode_all", "vidc_roc_decode_lists", "vidc_roc_decode_gather",
    "vidc_roc_decode_rows", "vidc_roc_last_decode_nonclean",
    "vidc_packed_bits_for", "vidc_packed_encode", "vidc_packed_destroy", "vidc_packed_compressed_bytes",
    "vidc_packed_bits", "vidc_packed_decode_all", "vidc_packed_decode_lists", "vidc_packed_decode_gather", "vidc_packed_get", "vidc_packed_export", "vidc_packed_total_words", "vidc_packed_export_all", "vidc_packed_import",
    "vidc_ef_encode", "vidc_ef_destroy", "vidc_ef_compressed_bytes", "vidc_ef_list_info", "vidc_ef_decode_all",
    "vidc_ef_get", "vidc_ef_perm", "vidc_ef_export", "vidc_ef_stream_words", "vidc_ef_export_all", "vidc_ef_import",
    "vidc_ef_encode_rows", "vidc_ef_decode_rows", "vidc_ef_decode_lists", "vidc_ef_decode_gather",
    "vidc_compact_rows_encode", "vidc_compact_destroy", "vidc_compact_bits", "vidc_compact_stride",
    "vidc_compact_size_in_bytes", "vidc_compact_rows_decode", "vidc_compact_export_row",
    "vidc_wt_build", "vidc_wt_destroy", "vidc_wt_size_in_bytes", "vidc_wt_levels", "vidc_wt_select",
    "vidc_wt_decode_all", "vidc_wt_decode_lists", "vidc_wt_decode_gather",
]


def lib():
    """The loaded C-ABI library (built on first use)."""
    global _lib
    if _lib is None:
        path = os.environ.get("VIDC_LIBRARY")  # a prebuilt libvidc.so somewhere else (A/B builds, packaging): loaded as is
        if not path:
            _build.build()  # returns at once unless the sources' content hash differs from the library's (build.py)
            path = LIB_PATH
        _lib = C.CDLL(path)
        _declare(_lib)
    return _lib


def check(status):
    if status != 0:
        raise VidcError(f"vidc status {status}: {lib().vidc_last_error().decode(errors='replace')}")


def ptr(x):
    """Raw address of a numpy array / torch tensor / int / None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        assert x.is_contiguous()
        return x.data_ptr()
    raise TypeError(type(x))


class Context:
    """vidc_ctx: one per (process, GPU).  Runs on torch's current stream when asked to."""

    def __init__(self, device=-1):
        h = _vp()
        check(lib().vidc_ctx_create(device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().vidc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        check(lib().vidc_ctx_set_stream(self.h, hip_stream))

    def synchronize(self):
        check(lib().vidc_ctx_synchronize(self.h))

    def trim(self):
        """Release the context's cached (idle) device / pinned blocks; returns the number of bytes freed."""
        freed = C.c_uint64(0)
        check(lib().vidc_ctx_trim(self.h, C.byref(freed)))
        return int(freed.value)

    def set_pool_poison(self, on):
        """Test aid: every cached device block is filled with 0xFF before it is handed out (vidc_ctx_debug_pool_poison)."""
        check(lib().vidc_ctx_debug_pool_poison(self.h, 1 if on else 0))

    def class_streams(self):
        """Streams the kernel classes of a large ROC call are spread over: 8 if the process started with GPU_MAX_HW_QUEUES >= 8, else 4."""
        return int(lib().vidc_ctx_class_streams(self.h))

    def d2h_bytes(self):
        """Id payload this context has copied device -> host so far (vidc_ctx_d2h_bytes)."""
        return int(lib().vidc_ctx_d2h_bytes(self.h))

    def last_kernel_ms(self):
        return float(lib().vidc_ctx_last_kernel_ms(self.h))

    def phase_ms(self, phase):
        """hipEvent time of one kernel phase of the latest call (VIDC_PHASE_* in include/vidc.h)."""
        return float(lib().vidc_ctx_phase_ms(self.h, phase))

    def chain_info(self, which):
        """What the chain launch (the kernel holding the longest lists) of the last ROC encode (0) / decode (1) processed."""
        ids, lists, longest, ub = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32()
        lib().vidc_ctx_chain_info(self.h, which, C.byref(ids), C.byref(lists), C.byref(longest), C.byref(ub))
        return dict(ids=ids.value, lists=lists.value, longest=longest.value, universe_bits=ub.value)


_default_ctx = {}


def default_context(device=None):
    """Context bound to the current torch device (created lazily, cached per device)."""
    import torch

    if not torch.cuda.is_available():
        raise VidcError("no HIP device: vector_db_id_compression_amd needs an MI355X (there is no CPU fallback)")
    if device is None:
        device = torch.cuda.current_device()
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    ctx = _default_ctx[device]
    # run on torch's CURRENT stream: tensors produced by torch ops are then ordered before the codec kernels
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    return ctx
