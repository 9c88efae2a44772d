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
