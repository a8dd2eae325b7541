"""ctypes binding of oracle/liblz4_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/lz4_oracle.h).  The product package never does.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblz4_oracle.so")

UNSUPPORTED_LEVEL = -1005


class Prefs(C.Structure):
    _fields_ = [
        ("block_size_id", C.c_uint32),
        ("block_mode", C.c_uint32),
        ("content_checksum", C.c_uint32),
        ("block_checksum", C.c_uint32),
        ("content_size", C.c_uint64),
        ("dict_id", C.c_uint32),
        ("compression_level", C.c_int32),
    ]


def build(force=False):
    src = os.path.join(_HERE, "lz4_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liblz4_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, sz, i64 = C.c_void_p, C.c_size_t, C.c_int64
        L.zo_compress_bound.restype = sz
        L.zo_compress_bound.argtypes = [sz]
        for name in ("zo_compress_default", "zo_decompress_safe", "zo_decompress_frame"):
            f = getattr(L, name)
            f.restype = i64
            f.argtypes = [u8p, sz, u8p, sz]
        L.zo_compress_fast.restype = i64
        L.zo_compress_fast.argtypes = [u8p, sz, u8p, sz, C.c_uint32]
        L.zo_compress_hc.restype = i64
        L.zo_compress_hc.argtypes = [u8p, sz, u8p, sz, C.c_int32]
        L.zo_sizeof_state.restype = sz
        L.zo_compress_fast_ext_state.restype = i64
        L.zo_compress_fast_ext_state.argtypes = [sz, u8p, sz, u8p, sz, C.c_uint32]
        L.zo_compress_dest_size.restype = i64
        L.zo_compress_dest_size.argtypes = [u8p, u8p, sz, C.POINTER(C.c_size_t)]
        L.zo_decompress_safe_partial.restype = i64
        L.zo_decompress_safe_partial.argtypes = [u8p, sz, u8p, sz, sz]
        L.zo_hc_reference_ub.restype = i64
        L.zo_hc_reference_ub.argtypes = [C.c_int]
        L.zo_xxh32.restype = C.c_uint32
        L.zo_xxh32.argtypes = [u8p, sz, C.c_uint32]
        L.zo_compress_frame_bound.restype = sz
        L.zo_compress_frame_bound.argtypes = [sz, C.POINTER(Prefs)]
        L.zo_compress_frame.restype = i64
        L.zo_compress_frame.argtypes = [u8p, sz, u8p, sz, C.POINTER(Prefs)]
        L.zo_header_size.restype = i64
        L.zo_header_size.argtypes = [u8p, sz]
        L.zo_batch_compress_default.restype = i64
        L.zo_batch_compress_default.argtypes = [u8p, sz, sz, u8p, sz, u8p]
        L.zo_batch_compress_hc.restype = i64
        L.zo_batch_compress_hc.argtypes = [u8p, sz, sz, u8p, sz, u8p, C.c_int32]
        L.zo_batch_decompress_safe.restype = i64
        L.zo_batch_decompress_safe.argtypes = [u8p, sz, u8p, sz, u8p, sz, u8p]
        _lib = L
    return _lib


def _buf(b):
    b = bytes(b)
    return (C.c_uint8 * max(1, len(b))).from_buffer_copy(b if b else b"\0"), len(b)


def compress_bound(n):
    return lib().zo_compress_bound(n)


def _call(fn, src, cap, *extra):
    s, n = _buf(src)
    d = (C.c_uint8 * max(1, cap))()
    r = fn(C.addressof(s), n, C.addressof(d), cap, *extra)
    if r < 0:
        return r
    return bytes(d[:r])


def compress_default(src, cap=None):
    cap = compress_bound(len(src)) if cap is None else cap
    return _call(lib().zo_compress_default, src, cap)


def compress_fast(src, accel, cap=None):
    cap = compress_bound(len(src)) if cap is None else cap
    return _call(lib().zo_compress_fast, src, cap, accel)


def compress_hc(src, level, cap=None):
    cap = compress_bound(len(src)) if cap is None else cap
    return _call(lib().zo_compress_hc, src, cap, level)


def hc_reference_ub(reset=True):
    """number of times compress_hc met the reference's u32 underflow (lz4hc.zig:636) since the last reset"""
    return lib().zo_hc_reference_ub(1 if reset else 0)


def decompress_safe(src, cap):
    return _call(lib().zo_decompress_safe, src, cap)


def compress_fast_ext_state(state_len, src, accel, cap=None):
    cap = compress_bound(len(src)) if cap is None else cap
    s, n = _buf(src)
    d = (C.c_uint8 * max(1, cap))()
    r = lib().zo_compress_fast_ext_state(state_len, C.addressof(s), n, C.addressof(d), cap, accel)
    return r if r < 0 else bytes(d[:r])


def compress_dest_size(src, cap):
    """-> (returned size or error, consumed source bytes)"""
    s, n = _buf(src)
    d = (C.c_uint8 * max(1, cap))()
    ss = C.c_size_t(n)
    r = lib().zo_compress_dest_size(C.addressof(s), C.addressof(d), cap, C.byref(ss))
    return r, ss.value


def decompress_safe_partial(src, cap, target):
    s, n = _buf(src)
    d = (C.c_uint8 * max(1, cap))()
    r = lib().zo_decompress_safe_partial(C.addressof(s), n, C.addressof(d), cap, target)
    return r if r < 0 else bytes(d[:r])


def xxh32(data, seed=0):
    s, n = _buf(data)
    return lib().zo_xxh32(C.addressof(s), n, seed)


def compress_frame_bound(n, prefs=None):
    return lib().zo_compress_frame_bound(n, C.byref(prefs) if prefs is not None else None)


def compress_frame(src, prefs=None, cap=None):
    cap = compress_frame_bound(len(src), prefs) if cap is None else cap
    return _call(lib().zo_compress_frame, src, cap, C.byref(prefs) if prefs is not None else None)


def decompress_frame(src, cap):
    return _call(lib().zo_decompress_frame, src, cap)


def header_size(src):
    s, n = _buf(src)
    return lib().zo_header_size(C.addressof(s), n)
