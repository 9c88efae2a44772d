"""Optional ctypes binding of the system liblz4 (1.9.3 in this image) — the C implementation
lz-fear imitates.  SECONDARY cross-check / yardstick only; tests skip when the
library is absent."""
import ctypes as C
import ctypes.util

_lib = None


def lib():
    global _lib
    if _lib is None:
        for name in ("liblz4.so.1", ctypes.util.find_library("lz4") or ""):
            if not name:
                continue
            try:
                L = C.CDLL(name)
            except OSError:
                continue
            L.LZ4_compress_default.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
            L.LZ4_createStream.restype = C.c_void_p
            L.LZ4_freeStream.argtypes = [C.c_void_p]
            L.LZ4_compress_fast_continue.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
            L.LZ4_decompress_safe.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
            L.LZ4_versionString.restype = C.c_char_p
            _lib = L
            break
        else:
            _lib = False
    return _lib or None


def available():
    return lib() is not None


def compress_default(data):
    data = bytes(data)
    cap = len(data) + len(data) // 255 + 64
    out = C.create_string_buffer(cap)
    n = lib().LZ4_compress_default(data, out, len(data), cap)
    return out.raw[:n]


def compress_fresh_stream(data):
    """LZ4_compress_fast_continue(accel=1) on a fresh LZ4_stream_t (hash5/byU32 for any size)."""
    data = bytes(data)
    cap = len(data) + len(data) // 255 + 64
    out = C.create_string_buffer(cap)
    st = lib().LZ4_createStream()
    n = lib().LZ4_compress_fast_continue(st, data, out, len(data), cap, 1)
    lib().LZ4_freeStream(st)
    return out.raw[:n]


def decompress_safe(data, max_out):
    data = bytes(data)
    out = C.create_string_buffer(max(max_out, 1))
    n = lib().LZ4_decompress_safe(data, out, len(data), max_out)
    return n, (out.raw[:n] if n >= 0 else b"")


# ---- LZ4F (the C frame layer), for interop checks of the frame format -------------------------------------------------
class _FrameInfo(C.Structure):
    _fields_ = [("blockSizeID", C.c_uint32), ("blockMode", C.c_uint32), ("contentChecksumFlag", C.c_uint32), ("frameType", C.c_uint32),
                ("contentSize", C.c_uint64), ("dictID", C.c_uint32), ("blockChecksumFlag", C.c_uint32)]


class _Preferences(C.Structure):
    _fields_ = [("frameInfo", _FrameInfo), ("compressionLevel", C.c_int), ("autoFlush", C.c_uint32), ("favorDecSpeed", C.c_uint32),
                ("reserved", C.c_uint32 * 3)]


def lz4f_compress(data, block_size_id=7, independent=True, content_checksum=True, block_checksums=False, level=0, content_size=False):
    """LZ4F_compressFrame (liblz4's own frame writer; level >= 3 is the HC encoder)."""
    L = lib()
    L.LZ4F_compressFrameBound.restype = C.c_size_t
    L.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.POINTER(_Preferences)]
    L.LZ4F_compressFrame.restype = C.c_size_t
    L.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(_Preferences)]
    L.LZ4F_isError.argtypes = [C.c_size_t]
    data = bytes(data)
    p = _Preferences()
    p.frameInfo.blockSizeID = block_size_id
    p.frameInfo.blockMode = 1 if independent else 0
    p.frameInfo.contentChecksumFlag = int(content_checksum)
    p.frameInfo.blockChecksumFlag = int(block_checksums)
    p.frameInfo.contentSize = len(data) if content_size else 0
    p.compressionLevel = level
    cap = L.LZ4F_compressFrameBound(len(data), C.byref(p))
    out = C.create_string_buffer(cap)
    n = L.LZ4F_compressFrame(out, cap, data, len(data), C.byref(p))
    assert not L.LZ4F_isError(n)
    return out.raw[:n]


def lz4f_decompress(frame, max_out):
    """LZ4F_decompress of one whole frame.  Returns (ok, bytes)."""
    L = lib()
    L.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    L.LZ4F_createDecompressionContext.restype = C.c_size_t
    L.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
    L.LZ4F_decompress.restype = C.c_size_t
    L.LZ4F_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_char_p, C.POINTER(C.c_size_t), C.c_void_p]
    L.LZ4F_isError.argtypes = [C.c_size_t]
    frame = bytes(frame)
    ctx = C.c_void_p()
    assert not L.LZ4F_isError(L.LZ4F_createDecompressionContext(C.byref(ctx), 100))
    out = C.create_string_buffer(max(max_out, 1))
    opos, ipos, ok = 0, 0, True
    while ipos < len(frame):
        dn = C.c_size_t(max_out - opos); sn = C.c_size_t(len(frame) - ipos)
        rc = L.LZ4F_decompress(ctx, C.byref(out, opos), C.byref(dn), frame[ipos:], C.byref(sn), None)
        if L.LZ4F_isError(rc):
            ok = False
            break
        opos += dn.value; ipos += sn.value
        if rc == 0:
            break
        if dn.value == 0 and sn.value == 0:
            ok = False
            break
    L.LZ4F_freeDecompressionContext(ctx)
    return ok, out.raw[:opos]
