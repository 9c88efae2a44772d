"""ctypes binding of oracle/liblzf_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The oracle is the CPU restatement of lz-fear (see oracle/lzf_oracle.h); the product path
never touches it.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

OK, UNEXPECTED_END, MEMORY_LIMIT_EXCEEDED, ZERO_DEDUP_OFFSET, INVALID_DEDUP_OFFSET = 0, 1, 2, 3, 4
OUTPUT_FULL, CONTRACT, OUT_CAPACITY = 5, 6, 7
F_INPUT_ERROR, F_WRONG_MAGIC, F_HEADER_CHECKSUM_FAIL, F_BLOCK_CHECKSUM_FAIL = 16, 17, 18, 19
F_FRAME_CHECKSUM_FAIL, F_BLOCK_LENGTH_OVERFLOW, F_BLOCK_SIZE_OVERFLOW = 20, 21, 22
F_UNIMPLEMENTED_BLOCKSIZE, F_UNSUPPORTED_VERSION, F_RESERVED_FLAG_BITS, F_RESERVED_BD_BITS = 23, 24, 25, 26
F_INVALID_BLOCK_SIZE, F_PANIC = 27, 28

STATUS_NAMES = {
    0: "Ok", 1: "UnexpectedEnd", 2: "MemoryLimitExceeded", 3: "ZeroDeduplicationOffset",
    4: "InvalidDeduplicationOffset", 5: "OutputFull", 6: "Contract", 7: "OutCapacity",
    16: "InputError", 17: "WrongMagic", 18: "HeaderChecksumFail", 19: "BlockChecksumFail",
    20: "FrameChecksumFail", 21: "BlockLengthOverflow", 22: "BlockSizeOverflow",
    23: "UnimplementedBlocksize", 24: "UnsupportedVersion", 25: "ReservedFlagBitsSet",
    26: "ReservedBdBitsSet", 27: "InvalidBlockSize", 28: "Panic",
}

TABLE_U32, TABLE_U16 = 0, 1


class U32Table(C.Structure):
    _fields_ = [("dict", C.c_uint32 * 4096), ("offset", C.c_uint64)]


class U16Table(C.Structure):
    _fields_ = [("dict", C.c_uint16 * 8192), ("offset", C.c_uint64)]


class Settings(C.Structure):
    _fields_ = [
        ("independent_blocks", C.c_int), ("block_checksums", C.c_int), ("content_checksum", C.c_int),
        ("block_size", C.c_uint64), ("dictionary", C.c_void_p), ("dictionary_len", C.c_uint64),
        ("has_dictionary_id", C.c_int), ("dictionary_id", C.c_uint32),
        ("has_content_size", C.c_int), ("content_size", C.c_uint64),
    ]


_lib = None


def build(force=False):
    so = os.path.join(ORACLE_DIR, "liblzf_oracle.so")
    src = os.path.join(ORACLE_DIR, "lzf_oracle.c")
    hdr = os.path.join(ORACLE_DIR, "lzf_oracle.h")
    stale = (not os.path.exists(so)) or any(
        os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(so) for f in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "liblzf_oracle.so"])
    return so


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.lzfo_compress2.restype = C.c_int
        L.lzfo_compress2.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.lzfo_decompress_raw.restype = C.c_int
        L.lzfo_decompress_raw.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                          C.c_void_p, C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t]
        L.lzfo_xxh32.restype = C.c_uint32
        L.lzfo_xxh32.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        L.lzfo_settings_default.argtypes = [C.POINTER(Settings)]
        L.lzfo_frame_compress.restype = C.c_int
        L.lzfo_frame_compress.argtypes = [C.POINTER(Settings), C.c_char_p, C.c_size_t,
                                          C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.lzfo_frame_compress_bound.restype = C.c_size_t
        L.lzfo_frame_compress_bound.argtypes = [C.POINTER(Settings), C.c_size_t]
        L.lzfo_frame_decompress.restype = C.c_int
        L.lzfo_frame_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                            C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def new_table(kind=TABLE_U32):
    return U32Table() if kind == TABLE_U32 else U16Table()


def compress2(data, cursor=0, kind=TABLE_U32, table=None, cap=None):
    """raw::compress2 (src/raw/compress/mod.rs:166).  Returns (status, bytes)."""
    data = bytes(data)
    if table is None:
        table = new_table(kind)
    if cap is None:
        cap = len(data) + len(data) // 255 + 64
    out = C.create_string_buffer(max(cap, 1))
    n = C.c_size_t(0)
    rc = lib().lzfo_compress2(data, len(data), cursor, kind, C.addressof(table), out, cap, C.byref(n))
    return rc, out.raw[: n.value]


def decompress_raw(data, prefix=b"", existing=b"", limit=None, cap=None):
    """raw::decompress_raw (src/raw/decompress.rs:59).  Returns (status, bytes incl. existing)."""
    data = bytes(data)
    prefix = bytes(prefix)
    if limit is None:
        limit = (1 << 63) - 1
    if cap is None:
        cap = len(existing) + min(limit, 1 << 26) + len(data) * 256 + 64
        cap = min(cap, 1 << 28)
    out = C.create_string_buffer(max(cap, 1))
    out[: len(existing)] = existing
    n = C.c_size_t(len(existing))
    rc = lib().lzfo_decompress_raw(data, len(data), prefix, len(prefix), out, C.byref(n), cap, limit)
    return rc, out.raw[: n.value]


def xxh32(data, seed=0):
    data = bytes(data)
    return lib().lzfo_xxh32(data, len(data), seed)


def make_settings(independent_blocks=True, block_checksums=False, content_checksum=True,
                  block_size=4 << 20, dictionary=None, dictionary_id=None, content_size=None,
                  dictionary_id_override=False):
    """CompressionSettings (src/framed/compress.rs:36-133).  dictionary_id follows the
    builder: dictionary(id, dict) sets both; dictionary_id_override=True + dictionary_id=None
    mirrors dictionary_id_nonsense_override(None)."""
    s = Settings()
    lib().lzfo_settings_default(C.byref(s))
    s.independent_blocks = int(independent_blocks)
    s.block_checksums = int(block_checksums)
    s.content_checksum = int(content_checksum)
    s.block_size = block_size
    keep = None
    if dictionary is not None:
        keep = C.create_string_buffer(bytes(dictionary), max(len(dictionary), 1))
        s.dictionary = C.cast(keep, C.c_void_p)
        s.dictionary_len = len(dictionary)
    if dictionary_id is not None:
        s.has_dictionary_id = 1
        s.dictionary_id = dictionary_id
    if content_size is not None:
        s.has_content_size = 1
        s.content_size = content_size
    s._keepalive = keep
    return s


def frame_compress(data, settings=None):
    data = bytes(data)
    if settings is None:
        settings = make_settings()
    cap = lib().lzfo_frame_compress_bound(C.byref(settings), len(data))
    out = C.create_string_buffer(max(cap, 1))
    n = C.c_size_t(0)
    rc = lib().lzfo_frame_compress(C.byref(settings), data, len(data), out, cap, C.byref(n))
    return rc, out.raw[: n.value]


def frame_decompress(data, dictionary=b"", cap=None):
    data = bytes(data)
    dictionary = bytes(dictionary)
    if cap is None:
        cap = max(1 << 20, len(data) * 300 + (8 << 20))
        cap = min(cap, 1 << 29)
    out = C.create_string_buffer(cap)
    n = C.c_size_t(0)
    used = C.c_size_t(0)
    rc = lib().lzfo_frame_decompress(data, len(data), dictionary, len(dictionary), out, cap,
                                     C.byref(n), C.byref(used))
    return rc, out.raw[: n.value], used.value
