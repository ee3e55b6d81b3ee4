"""ctypes binding of include/lzfse_mi.h. Loading fails loudly when the HIP library is missing."""
import ctypes as C
import os

from . import build as _build

MAX_STAGES = 24


class Timings(C.Structure):
    _fields_ = [("n_stages", C.c_int), ("names", C.c_char_p * MAX_STAGES), ("ms", C.c_float * MAX_STAGES),
                ("launches", C.c_uint64 * MAX_STAGES)]


WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)   # lzfse_mi_write_fn

_lib = None
_diag_lib = None

_SYMBOLS = [
    "lzfse_mi_device_count", "lzfse_mi_create", "lzfse_mi_destroy", "lzfse_mi_status_string", "lzfse_mi_version", "lzfse_mi_set_stream",
    "lzfse_mi_encode_bound", "lzfse_mi_encode", "lzfse_mi_decode", "lzfse_mi_decode_size",
    "lzfse_mi_encode_batch", "lzfse_mi_decode_batch", "lzfse_mi_decode_batch_device",
    "lzfse_mi_encode_batch_device", "lzfse_mi_enable_timing", "lzfse_mi_get_timings", "lzfse_mi_encode_small",
    "lzfse_mi_last_error_detail", "lzfse_mi_set_option", "lzfse_mi_chunked_bound", "lzfse_mi_encode_chunked",
    "lzfse_mi_decode_chunked_size", "lzfse_mi_decode_chunked", "lzfse_mi_dstream_create", "lzfse_mi_dstream_feed", "lzfse_mi_dstream_reserve", "lzfse_mi_dstream_commit",
    "lzfse_mi_dstream_totals", "lzfse_mi_dstream_destroy", "lzfse_mi_decode_headroom",
    "lzfse_mi_encode_ring", "lzfse_mi_encode_ring_batch", "lzfse_mi_encode_ring_batch_device",
    "lzfse_mi_get_info", "lzfse_mi_estream_create", "lzfse_mi_estream_feed", "lzfse_mi_estream_reserve", "lzfse_mi_estream_commit", "lzfse_mi_estream_finish", "lzfse_mi_estream_destroy",
]


def lib(diag=False):
    """The product library; diag=True: the diagnostic build (LZFSE_MI_OPT_DIAG_* options, debug hook) for tests."""
    global _lib, _diag_lib
    if diag:
        if _diag_lib is None:
            _diag_lib = _load(_build.DIAG_LIB_PATH)
        return _diag_lib
    if _lib is None:
        _lib = _load(_build.LIB_PATH)
    return _lib


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`. "
                           "There is no CPU fallback for the MI355X codec.")
    L = C.CDLL(path)
    for s in _SYMBOLS:
        getattr(L, s)  # raises AttributeError if the ABI is incomplete
    vp, sz, u64p, ip = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_int)
    L.lzfse_mi_device_count.restype = C.c_int
    L.lzfse_mi_device_count.argtypes = []
    L.lzfse_mi_create.restype = C.c_int
    L.lzfse_mi_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.lzfse_mi_destroy.restype = None
    L.lzfse_mi_destroy.argtypes = [vp]
    L.lzfse_mi_status_string.restype = C.c_char_p
    L.lzfse_mi_status_string.argtypes = [C.c_int]
    L.lzfse_mi_version.restype = C.c_char_p
    L.lzfse_mi_set_stream.restype = C.c_int
    L.lzfse_mi_set_stream.argtypes = [vp, vp]
    L.lzfse_mi_encode_bound.restype = sz
    L.lzfse_mi_encode_bound.argtypes = [sz]
    for name in ("lzfse_mi_encode", "lzfse_mi_decode", "lzfse_mi_encode_ring"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
    L.lzfse_mi_encode_small.restype = C.c_int
    L.lzfse_mi_encode_small.argtypes = [vp, sz, vp, sz, C.POINTER(sz)]
    L.lzfse_mi_decode_size.restype = C.c_int
    L.lzfse_mi_decode_size.argtypes = [vp, sz, u64p]
    for name in ("lzfse_mi_encode_batch", "lzfse_mi_decode_batch", "lzfse_mi_encode_ring_batch"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), ip]
    for name in ("lzfse_mi_encode_batch_device", "lzfse_mi_decode_batch_device", "lzfse_mi_encode_ring_batch_device"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [vp, sz, vp, u64p, u64p, vp, u64p, u64p, u64p, ip]
    L.lzfse_mi_last_error_detail.restype = C.c_int
    L.lzfse_mi_last_error_detail.argtypes = [vp, sz, C.POINTER(C.c_uint32)]
    L.lzfse_mi_enable_timing.restype = C.c_int
    L.lzfse_mi_enable_timing.argtypes = [vp, C.c_int]
    L.lzfse_mi_get_timings.restype = C.c_int
    L.lzfse_mi_get_timings.argtypes = [vp, C.POINTER(Timings)]
    L.lzfse_mi_chunked_bound.restype = sz
    L.lzfse_mi_chunked_bound.argtypes = [sz, sz]
    L.lzfse_mi_encode_chunked.restype = C.c_int
    L.lzfse_mi_encode_chunked.argtypes = [C.POINTER(vp), C.c_int, vp, sz, sz, vp, sz, C.POINTER(sz)]
    L.lzfse_mi_decode_chunked_size.restype = C.c_int
    L.lzfse_mi_decode_chunked_size.argtypes = [vp, sz, u64p]
    L.lzfse_mi_decode_chunked.restype = C.c_int
    L.lzfse_mi_decode_chunked.argtypes = [C.POINTER(vp), C.c_int, vp, sz, vp, sz, C.POINTER(sz)]
    L.lzfse_mi_decode_headroom.restype = sz
    L.lzfse_mi_decode_headroom.argtypes = [vp, sz]
    L.lzfse_mi_dstream_create.restype = C.c_int
    L.lzfse_mi_dstream_create.argtypes = [vp, sz, C.POINTER(vp)]
    L.lzfse_mi_dstream_feed.restype = C.c_int
    L.lzfse_mi_dstream_feed.argtypes = [vp, vp, sz, C.c_int, WRITE_FN, vp]
    L.lzfse_mi_dstream_reserve.restype = C.c_int
    L.lzfse_mi_dstream_reserve.argtypes = [vp, sz, C.POINTER(vp)]
    L.lzfse_mi_dstream_commit.restype = C.c_int
    L.lzfse_mi_dstream_commit.argtypes = [vp, sz, C.c_int, WRITE_FN, vp]
    L.lzfse_mi_dstream_totals.restype = C.c_int
    L.lzfse_mi_dstream_totals.argtypes = [vp, u64p, u64p]
    L.lzfse_mi_dstream_destroy.restype = None
    L.lzfse_mi_dstream_destroy.argtypes = [vp]
    L.lzfse_mi_estream_create.restype = C.c_int
    L.lzfse_mi_estream_create.argtypes = [vp, sz, C.POINTER(vp)]
    L.lzfse_mi_estream_feed.restype = C.c_int
    L.lzfse_mi_estream_feed.argtypes = [vp, vp, sz, WRITE_FN, vp]
    L.lzfse_mi_estream_reserve.restype = C.c_int
    L.lzfse_mi_estream_reserve.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(sz), WRITE_FN, vp]
    L.lzfse_mi_estream_commit.restype = C.c_int
    L.lzfse_mi_estream_commit.argtypes = [vp, sz]
    L.lzfse_mi_estream_finish.restype = C.c_int
    L.lzfse_mi_estream_finish.argtypes = [vp, WRITE_FN, vp, u64p, u64p]
    L.lzfse_mi_estream_destroy.restype = None
    L.lzfse_mi_estream_destroy.argtypes = [vp]
    L.lzfse_mi_get_info.restype = C.c_int
    L.lzfse_mi_get_info.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
    L.lzfse_mi_set_option.restype = C.c_int
    L.lzfse_mi_set_option.argtypes = [vp, C.c_int, C.c_int64]
    return L
