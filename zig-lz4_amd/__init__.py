"""zig_lz4_amd -- Python view of the MI355X-native LZ4 codec's C ABI (include/zlz4_amd.h).

The product is the HIP shared library `libzlz4_amd.so` built from csrc/; this module is
the thin ctypes layer the tests and bench.py use.  It mirrors the names of the reference's
public facade (src/root.zig:1-57): compressBound / compressDefault / compressFast /
compressHC / decompressSafe and the `lz4f` namespace, with the reference's error
behaviour (Zig error unions -> Python exceptions carrying the same error name).

There is no CPU implementation here: if the library is missing, importing any compute
entry point raises; if no gfx950 device is present the calls raise Lz4Error("DeviceError").
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZLZ4_AMD_LIB selects a diagnostic build of the same library (e.g. the -DZLZ4_STAMPS one); never a fallback
LIB_PATH = os.environ.get("ZLZ4_AMD_LIB") or os.path.join(_HERE, "libzlz4_amd.so")

# constants re-exported by src/root.zig:46-49 and src/lz4hc.zig:28-31
MINMATCH = 4
LZ4_MAX_INPUT_SIZE = 0x7E000000
LZ4_DISTANCE_MAX = 65535
LZ4HC_CLEVEL_MIN = 2
LZ4HC_CLEVEL_DEFAULT = 9
LZ4HC_CLEVEL_MAX = 12

ERR_DEVICE = -7
ERR_UNSUPPORTED = -8


class Lz4Error(Exception):
    """Mirror of lz4.Error / lz4f.Error (src/lz4.zig:48-55, src/lz4f.zig:31-55)."""

    def __init__(self, code, name):
        super().__init__("%s (%d)" % (name, code))
        self.code = code
        self.name = name


class Prefs(C.Structure):
    """zlz4f_prefs == lz4f.Preferences + FrameInfo flattened (src/lz4f.zig:106-122)."""
    _fields_ = [
        ("block_size_id", C.c_uint32),
        ("block_mode", C.c_uint32),
        ("content_checksum", C.c_uint32),
        ("block_checksum", C.c_uint32),
        ("content_size", C.c_uint64),
        ("dict_id", C.c_uint32),
        ("compression_level", C.c_int32),
    ]


# every symbol include/zlz4_amd.h declares: name -> (restype, argtypes)
_VP, _SZ, _I64, _I32, _U32 = C.c_void_p, C.c_size_t, C.c_int64, C.c_int32, C.c_uint32
_PP = C.POINTER(Prefs)
SYMBOLS = {
    "zlz4_compress_bound": (_SZ, [_SZ]),
    "zlz4_compress_default": (_I64, [_VP, _SZ, _VP, _SZ]),
    "zlz4_compress_fast": (_I64, [_VP, _SZ, _VP, _SZ, _U32]),
    "zlz4_compress_hc": (_I64, [_VP, _SZ, _VP, _SZ, _I32]),
    "zlz4_sizeof_state_hc": (_SZ, []),
    "zlz4_compress_hc_ext_state": (_I64, [_VP, _SZ, _VP, _SZ, _VP, _SZ, _I32]),
    "zlz4_decompress_safe": (_I64, [_VP, _SZ, _VP, _SZ]),
    "zlz4_decompress_safe_partial": (_I64, [_VP, _SZ, _VP, _SZ, _SZ]),
    "zlz4_sizeof_state": (_SZ, []),
    "zlz4_compress_fast_ext_state": (_I64, [_VP, _SZ, _VP, _SZ, _VP, _SZ, _U32]),
    "zlz4_compress_dest_size": (_I64, [_VP, _VP, _SZ, C.POINTER(C.c_size_t)]),
    "zlz4_batch_compress_fast": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _U32, _U32, _U32]),
    "zlz4_batch_decompress_safe": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _U32]),
    "zlz4_batch_compress_hc_workspace": (_SZ, [_U32, _U32]),
    "zlz4_batch_compress_hc": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _U32, _U32, _I32, _VP, _SZ]),
    "zlz4_batch_verify": (_I64, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _U32]),
    "zlz4f_compress_frame_bound": (_SZ, [_SZ, _PP]),
    "zlz4f_compress_frame": (_I64, [_VP, _SZ, _VP, _SZ, _PP]),
    "zlz4f_decompress_frame": (_I64, [_VP, _SZ, _VP, _SZ]),
    "zlz4f_header_size": (_I64, [_VP, _SZ]),
    "zlz4f_compress_frame_device": (_I64, [_VP, _VP, _SZ, _VP, _SZ, _PP]),
    "zlz4f_decompress_frame_device": (_I64, [_VP, _VP, _SZ, _VP, _SZ]),
    "zlz4f_compress_frame_segment_device": (_I64, [_VP, _VP, _SZ, _VP, _SZ, _PP, _U32]),
    "zlz4f_decompress_frame_segment_device": (_I64, [_VP, _VP, _SZ, _VP, _SZ, _PP, _U32]),
    "zlz4_device_check": (_I32, []),
    "zlz4_version_string": (C.c_char_p, []),
    "zlz4_error_name": (C.c_char_p, [_I64]),
    "zlz4_release_device_cache": (None, []),
}

_lib = None


def lib():
    """Load libzlz4_amd.so (fails loudly: no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `make` (hipcc --offload-arch=gfx950) -- "
                "there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7).
        # If torch is going to be used in this process it must be loaded first, so that our library's
        # DT_NEEDED libamdhip64.so.7 binds to the same runtime (device pointers / streams are shared).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)      # AttributeError if the library does not export it
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def error_name(code):
    return lib().zlz4_error_name(code).decode()


def _check(r):
    if r < 0:
        raise Lz4Error(r, error_name(r))
    return r


def device_available():
    return lib().zlz4_device_check() == 0


# ----------------------------------------------------------------------------- root.zig names
def compressBound(input_size):
    """lz4.compressBound, src/lz4.zig:80-83."""
    return lib().zlz4_compress_bound(input_size)


def _in(b):
    b = bytes(b)
    buf = (C.c_uint8 * max(1, len(b))).from_buffer_copy(b if b else b"\0")
    return buf, len(b)


def _run(fn, src, cap, *extra):
    s, n = _in(src)
    d = (C.c_uint8 * max(1, cap))()
    r = _check(fn(C.addressof(s), n, C.addressof(d), cap, *extra))
    return bytes(d[:r])


def compressDefault(src, dst_cap=None):
    """lz4.compressDefault(src, dst), src/lz4.zig:283-285; dst_cap defaults to compressBound(len(src))."""
    cap = compressBound(len(src)) if dst_cap is None else dst_cap
    return _run(lib().zlz4_compress_default, src, cap)


def compressFast(src, acceleration, dst_cap=None):
    """lz4.compressFast(src, dst, acceleration), src/lz4.zig:292-447."""
    cap = compressBound(len(src)) if dst_cap is None else dst_cap
    return _run(lib().zlz4_compress_fast, src, cap, acceleration)


def compressHC(src, compression_level, dst_cap=None):
    """lz4hc.compressHC(src, dst, level), src/lz4hc.zig:1440-1453."""
    cap = compressBound(len(src)) if dst_cap is None else dst_cap
    return _run(lib().zlz4_compress_hc, src, cap, compression_level)


def sizeofStateHC():
    """lz4hc.sizeofStateHC, src/lz4hc.zig:1492-1494."""
    return lib().zlz4_sizeof_state_hc()


def compressHCExtState(state_len, src, compression_level, dst_cap=None):
    """lz4hc.compressHCExtState(ctx, src, dst, level), src/lz4hc.zig:1457-1489 (fresh context, given by its size)."""
    cap = compressBound(len(src)) if dst_cap is None else dst_cap
    s, n = _in(src)
    st = (C.c_uint8 * max(1, state_len))()
    d = (C.c_uint8 * max(1, cap))()
    r = _check(lib().zlz4_compress_hc_ext_state(C.addressof(st), state_len, C.addressof(s), n, C.addressof(d), cap, compression_level))
    return bytes(d[:r])


def decompressSafe(src, dst_cap):
    """lz4.decompressSafe(src, dst), src/lz4.zig:257-259; dst_cap == dst.len."""
    return _run(lib().zlz4_decompress_safe, src, dst_cap)


def decompressSafePartial(src, dst_cap, target_output_size):
    """lz4.decompressSafePartial(src, dst, targetOutputSize), src/lz4.zig:619-621."""
    return _run(lib().zlz4_decompress_safe_partial, src, dst_cap, target_output_size)


def sizeofState():
    """lz4.sizeofState, src/lz4.zig:524-526."""
    return lib().zlz4_sizeof_state()


def compressFastExtState(state_len, src, acceleration, dst_cap=None):
    """lz4.compressFastExtState(state, src, dst, acceleration), src/lz4.zig:531-546 (state given by its length)."""
    cap = compressBound(len(src)) if dst_cap is None else dst_cap
    s, n = _in(src)
    st = (C.c_uint8 * max(1, state_len))()
    d = (C.c_uint8 * max(1, cap))()
    r = _check(lib().zlz4_compress_fast_ext_state(C.addressof(st), state_len, C.addressof(s), n, C.addressof(d), cap, acceleration))
    return bytes(d[:r])


def compressDestSize(src, dst_cap):
    """lz4.compressDestSize(src, dst, &srcSize), src/lz4.zig:551-616 -> (compressed bytes, consumed source bytes)."""
    s, n = _in(src)
    d = (C.c_uint8 * max(1, dst_cap))()
    ss = C.c_size_t(n)
    r = _check(lib().zlz4_compress_dest_size(C.addressof(s), C.addressof(d), dst_cap, C.byref(ss)))
    return bytes(d[:r]), ss.value


class lz4f:
    """Mirror of the `lz4f` namespace (src/root.zig:54-55, src/lz4f.zig)."""
    MAGICNUMBER = 0x184D2204
    Preferences = Prefs

    @staticmethod
    def compressFrameBound(src_size, prefs=None):
        return lib().zlz4f_compress_frame_bound(src_size, C.byref(prefs) if prefs is not None else None)

    @staticmethod
    def compressFrame(src, prefs=None, dst_cap=None):
        cap = lz4f.compressFrameBound(len(src), prefs) if dst_cap is None else dst_cap
        return _run(lib().zlz4f_compress_frame, src, cap, C.byref(prefs) if prefs is not None else None)

    @staticmethod
    def decompressFrame(src, dst_cap):
        return _run(lib().zlz4f_decompress_frame, src, dst_cap)

    @staticmethod
    def headerSize(src):
        s, n = _in(src)
        return _check(lib().zlz4f_header_size(C.addressof(s), n))

    # device-resident variants (torch CUDA uint8 tensors in, frame / content size out)
    @staticmethod
    def compressFrameDevice(d_src, d_dst, prefs=None):
        return _check(lib().zlz4f_compress_frame_device(_stream(), _ptr(d_src), d_src.numel(), _ptr(d_dst), d_dst.numel(),
                                                        C.byref(prefs) if prefs is not None else None))

    @staticmethod
    def decompressFrameDevice(d_frame, frame_len, d_dst):
        return _check(lib().zlz4f_decompress_frame_device(_stream(), _ptr(d_frame), frame_len, _ptr(d_dst), d_dst.numel()))

    SEG_FIRST, SEG_LAST = 1, 2

    @staticmethod
    def compressFrameSegmentDevice(d_src, d_dst, prefs, seg_flags):
        """One rank's block segment of a frame split over several GPUs (include/zlz4_amd.h)."""
        return _check(lib().zlz4f_compress_frame_segment_device(_stream(), _ptr(d_src), d_src.numel(), _ptr(d_dst),
                                                                d_dst.numel(), C.byref(prefs) if prefs is not None else None,
                                                                seg_flags))

    @staticmethod
    def decompressFrameSegmentDevice(d_seg, seg_len, d_dst, prefs, seg_flags):
        return _check(lib().zlz4f_decompress_frame_segment_device(_stream(), _ptr(d_seg), seg_len, _ptr(d_dst), d_dst.numel(),
                                                                  C.byref(prefs) if prefs is not None else None, seg_flags))


# ----------------------------------------------------------------------------- batch (device pointers)
def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def batch_compress_fast(d_in, in_off, in_len, d_out, out_off, out_cap, result, max_in_len, acceleration=1):
    """zlz4_batch_compress_fast on torch CUDA tensors (uint8 / int64 offsets / int32 lengths / int64 result)."""
    _check(lib().zlz4_batch_compress_fast(_stream(), _ptr(d_in), _ptr(in_off), _ptr(in_len), _ptr(d_out),
                                          _ptr(out_off), _ptr(out_cap), _ptr(result), in_len.numel(),
                                          max_in_len, acceleration))


def batch_decompress_safe(d_in, in_off, in_len, d_out, out_off, out_cap, result):
    _check(lib().zlz4_batch_decompress_safe(_stream(), _ptr(d_in), _ptr(in_off), _ptr(in_len), _ptr(d_out),
                                            _ptr(out_off), _ptr(out_cap), _ptr(result), in_len.numel()))


def batch_compress_hc_workspace(nblocks, max_in_len):
    return lib().zlz4_batch_compress_hc_workspace(nblocks, max_in_len)


def batch_compress_hc(d_in, in_off, in_len, d_out, out_off, out_cap, result, max_in_len, level, workspace):
    _check(lib().zlz4_batch_compress_hc(_stream(), _ptr(d_in), _ptr(in_off), _ptr(in_len), _ptr(d_out),
                                        _ptr(out_off), _ptr(out_cap), _ptr(result), in_len.numel(), max_in_len,
                                        level, _ptr(workspace), workspace.numel()))


def batch_verify(d_in, in_off, in_len, d_comp, comp_off, comp_result, verify):
    """zlz4_batch_verify: decode every compressed block on the device and compare with its input; returns the number of
    blocks that do not round-trip (verify[i] = the compress result, or ZLZ4_ERR_VERIFY = -9)."""
    return _check(lib().zlz4_batch_verify(_stream(), _ptr(d_in), _ptr(in_off), _ptr(in_len), _ptr(d_comp), _ptr(comp_off),
                                          _ptr(comp_result), _ptr(verify), in_len.numel()))
