"""ctypes binding of include/lzfear_hip.h (liblzfear_hip.so).

Loading fails loudly if the library has not been built; calling any codec entry point fails
loudly (LzfError) when there is no HIP device — there is no CPU fallback.
"""
import ctypes as C
import os

from . import build as _build

OK, UNEXPECTED_END, MEMORY_LIMIT_EXCEEDED, ZERO_DEDUP_OFFSET, INVALID_DEDUP_OFFSET = 0, 1, 2, 3, 4
OUTPUT_FULL, CONTRACT, OUT_CAPACITY = 5, 6, 7
E_NO_DEVICE, E_HIP, E_INVALID = -1, -2, -3
TABLE_U32, TABLE_U16 = 0, 1
KINDS_U32, KINDS_U16, KINDS_U32_FRESH_ONLY = 1, 2, 4
E_NO_MEMORY = -4
CJOB_TABLE_READONLY = 1


class LzfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lzfear_hip error {code}: {msg}")
        self.code = code


class U32Table(C.Structure):
    """lzf_u32_table — src/raw/compress/mod.rs:27-36"""
    _fields_ = [("dict", C.c_uint32 * 4096), ("offset", C.c_uint64)]


class U16Table(C.Structure):
    """lzf_u16_table — src/raw/compress/mod.rs:78-87"""
    _fields_ = [("dict", C.c_uint16 * 8192), ("offset", C.c_uint64)]


class CompressJob(C.Structure):
    _fields_ = [("input", C.c_void_p), ("input_len", C.c_uint64), ("cursor", C.c_uint64),
                ("out", C.c_void_p), ("out_cap", C.c_uint64), ("table", C.c_void_p),
                ("table_kind", C.c_uint32), ("flags", C.c_uint32)]


class DecompressJob(C.Structure):
    _fields_ = [("input", C.c_void_p), ("input_len", C.c_uint64), ("prefix", C.c_void_p),
                ("prefix_len", C.c_uint64), ("out", C.c_void_p), ("out_existing_len", C.c_uint64),
                ("out_cap", C.c_uint64), ("output_limit", C.c_uint64)]


class JobResult(C.Structure):
    _fields_ = [("out_len", C.c_uint64), ("status", C.c_int32), ("reserved", C.c_uint32)]


assert C.sizeof(CompressJob) == 56 and C.sizeof(DecompressJob) == 64 and C.sizeof(JobResult) == 16

class Settings(C.Structure):
    """lzf_settings — CompressionSettings, src/framed/compress.rs:36-55"""
    _fields_ = [("independent_blocks", C.c_int32), ("block_checksums", C.c_int32), ("content_checksum", C.c_int32),
                ("has_dictionary_id", C.c_int32), ("block_size", C.c_uint64), ("dictionary", C.c_void_p),
                ("dictionary_len", C.c_uint64), ("dictionary_id", C.c_uint32), ("has_content_size", C.c_int32),
                ("content_size", C.c_uint64)]


class FrameInfo(C.Structure):
    _fields_ = [("flags", C.c_uint8), ("bd", C.c_uint8), ("header_len", C.c_uint16), ("dictionary_id", C.c_uint32),
                ("has_dictionary_id", C.c_int32), ("has_content_size", C.c_int32), ("content_size", C.c_uint64),
                ("block_maxsize", C.c_uint64)]


EXPORTS = [
    "lzf_abi_version", "lzf_last_error", "lzf_device_count", "lzf_compress_batch",
    "lzf_decompress_batch", "lzf_decompress_batch_sized", "lzf_last_decompress_launch", "lzf_last_compress_launch", "lzf_table_replace_host", "lzf_table_offset_host", "lzf_compress2_host_writer", "lzf_table_seed_from_dictionary", "lzf_table_offset", "lzf_table_offset_batch",
    "lzf_chain_decompress_step",
    "lzf_xxh32_batch", "lzf_copy_ranges", "lzf_compress_batch_host", "lzf_decompress_batch_host", "lzf_xxh32_batch_host",
]
FRAME_EXPORTS = [
    "lzf_settings_default", "lzf_frame_compress_bound", "lzf_frame_compress", "lzf_frame_read_header",
    "lzf_frame_decompress", "lzf_xxh32", "lzf_frame_assemble", "lzf_frame_compress_many", "lzf_frame_decompress_many",
    "lzf_xxh32_reset", "lzf_xxh32_update", "lzf_xxh32_digest",
    "lzf_frame_reader_new", "lzf_frame_reader_free", "lzf_frame_reader_info", "lzf_frame_reader_decode_block",
    "lzf_frame_reader_finished", "lzf_frame_reader_consumed",
    "lzf_frame_writer_new", "lzf_frame_writer_write", "lzf_frame_writer_finish", "lzf_frame_writer_sink_error", "lzf_frame_writer_free",
    "lzf_frame_get_stats", "lzf_frame_release_scratch", "lzf_frame_set_host_threads", "lzf_frame_set_memory_budget",
    "lzf_frame_set_pinned_limit",
]


class FrameStats(C.Structure):
    """lzf_frame_stats"""
    _fields_ = [(n, C.c_uint64) for n in ("calls", "device_block_hashes", "host_block_hashes", "device_content_hashes",
                                          "host_content_hashes", "h2d_copies", "d2h_copies", "h2d_bytes", "d2h_bytes",
                                          "pinned_bytes")]


class Xxh32State(C.Structure):
    _fields_ = [("v", C.c_uint32 * 4), ("buf", C.c_uint8 * 16), ("fill", C.c_uint32), ("seed", C.c_uint32), ("total", C.c_uint64)]

_lib = None


def lib_path():
    return _build.LIB_PATH


def lib():
    """Load liblzfear_hip.so (must already be built: __graft_entry__.build())."""
    global _lib
    if _lib is None:
        try:
            # One HIP runtime per process: when torch is installed, load its ROCm libraries
            # first so that liblzfear_hip.so binds to the same libamdhip64 torch uses.
            import torch  # noqa: F401
        except ImportError:
            pass
        path = lib_path()
        if not os.path.exists(path):
            raise LzfError(E_INVALID, f"{path} is missing — run __graft_entry__.build() "
                                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(path)
        L.lzf_last_error.restype = C.c_char_p
        L.lzf_last_decompress_launch.restype = C.c_char_p
        L.lzf_last_compress_launch.restype = C.c_char_p
        L.lzf_compress_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.lzf_decompress_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.lzf_decompress_batch_sized.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        L.lzf_table_seed_from_dictionary.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.lzf_table_offset.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        L.lzf_xxh32_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.lzf_copy_ranges.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        L.lzf_compress_batch_host.argtypes = [C.POINTER(CompressJob), C.POINTER(JobResult), C.c_uint32]
        L.lzf_decompress_batch_host.argtypes = [C.POINTER(DecompressJob), C.POINTER(JobResult), C.c_uint32]
        L.lzf_xxh32_batch_host.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint32]
        L.lzf_settings_default.argtypes = [C.POINTER(Settings)]
        L.lzf_frame_compress_bound.restype = C.c_size_t
        L.lzf_frame_compress_bound.argtypes = [C.POINTER(Settings), C.c_size_t]
        L.lzf_frame_compress.argtypes = [C.POINTER(Settings), C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.lzf_frame_read_header.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(FrameInfo)]
        L.lzf_frame_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                           C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.lzf_frame_compress_many.argtypes = [C.POINTER(Settings), C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t),
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.lzf_frame_decompress_many.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t,
                                                C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                                C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.lzf_table_offset_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.lzf_chain_decompress_step.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lzf_xxh32.restype = C.c_uint32
        L.lzf_xxh32.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        L.lzf_xxh32_reset.argtypes = [C.POINTER(Xxh32State), C.c_uint32]
        L.lzf_xxh32_reset.restype = None
        L.lzf_xxh32_update.argtypes = [C.POINTER(Xxh32State), C.c_char_p, C.c_size_t]
        L.lzf_xxh32_update.restype = None
        L.lzf_xxh32_digest.argtypes = [C.POINTER(Xxh32State)]
        L.lzf_xxh32_digest.restype = C.c_uint32
        L.lzf_frame_assemble.argtypes = [C.POINTER(Settings), C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.lzf_frame_reader_new.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.lzf_frame_reader_free.argtypes = [C.c_void_p]
        L.lzf_frame_reader_free.restype = None
        L.lzf_frame_reader_info.argtypes = [C.c_void_p, C.POINTER(FrameInfo)]
        L.lzf_frame_reader_info.restype = None
        L.lzf_frame_reader_decode_block.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.lzf_frame_reader_finished.argtypes = [C.c_void_p]
        L.lzf_frame_reader_consumed.argtypes = [C.c_void_p]
        L.lzf_frame_reader_consumed.restype = C.c_size_t
        L.lzf_frame_get_stats.argtypes = [C.POINTER(FrameStats)]
        L.lzf_frame_get_stats.restype = None
        L.lzf_frame_release_scratch.restype = None
        L.lzf_frame_set_host_threads.argtypes = [C.c_uint32]
        L.lzf_frame_set_host_threads.restype = None
        L.lzf_frame_set_memory_budget.argtypes = [C.c_size_t]
        L.lzf_frame_set_memory_budget.restype = None
        L.lzf_frame_set_pinned_limit.argtypes = [C.c_size_t]
        L.lzf_frame_set_pinned_limit.restype = None
        _lib = L
    return _lib


def frame_stats():
    st = FrameStats()
    lib().lzf_frame_get_stats(C.byref(st))
    return {n: getattr(st, n) for n, _ in FrameStats._fields_}


def check(rc):
    if rc < 0:
        raise LzfError(rc, lib().lzf_last_error().decode())
    return rc


def device_count():
    return check(lib().lzf_device_count())


# ---- host-buffer batch helpers (bytes in, bytes out) --------------------------------------

def xxh32_blocks_host(blocks):
    """XXH32 (seed 0) of every bytes object in `blocks`, computed on the device (lzf_xxh32_batch_host)."""
    blocks = [bytes(b) for b in blocks]
    n = len(blocks)
    if n == 0:
        return []
    ptrs = (C.c_char_p * n)(*blocks)
    lens = (C.c_uint64 * n)(*[len(b) for b in blocks])
    out = (C.c_uint32 * n)()
    check(lib().lzf_xxh32_batch_host(ptrs, lens, out, n))
    return list(out)


def _bytes_address(b):
    """Address of a bytes object's payload (read-only use; the caller keeps `b` alive).  No copy, and no ctypes.cast: cast() objects
    reference themselves, so buffers hanging off them live until the cycle collector runs."""
    cp = C.c_char_p(b)                            # (holds a reference to b and, in its own storage, the pointer)
    return C.c_void_p.from_address(C.addressof(cp)).value


def compress_blocks_host(items):
    """items: list of dict(input=bytes, cursor=int, kind=TABLE_*, table=None|U32Table|U16Table,
    out_cap=int, readonly=bool).  Returns list of (status, bytes)."""
    n = len(items)
    if n == 0:
        return []
    jobs = (CompressJob * n)()
    res = (JobResult * n)()
    keep = []
    for i, it in enumerate(items):
        data = bytes(it["input"])
        cap = it.get("out_cap")
        if cap is None:
            cap = len(data) + len(data) // 255 + 64
        ib = C.create_string_buffer(data, max(len(data), 1))
        ob = C.create_string_buffer(max(cap, 1))
        keep.append((ib, ob))
        jobs[i].input = C.addressof(ib)
        jobs[i].input_len = len(data)
        jobs[i].cursor = it.get("cursor", 0)
        jobs[i].out = C.addressof(ob)
        jobs[i].out_cap = cap
        t = it.get("table")
        jobs[i].table = C.addressof(t) if t is not None else None
        jobs[i].table_kind = it.get("kind", TABLE_U32)
        jobs[i].flags = CJOB_TABLE_READONLY if it.get("readonly") else 0
    check(lib().lzf_compress_batch_host(jobs, res, n))
    return [(res[i].status, C.string_at(keep[i][1], res[i].out_len) if res[i].status == OK else b"") for i in range(n)]


def decompress_blocks_host(items):
    """items: list of dict(input=bytes, prefix=bytes, existing=bytes, limit=int, out_cap=int).
    Returns list of (status, bytes incl. existing)."""
    n = len(items)
    if n == 0:
        return []
    jobs = (DecompressJob * n)()
    res = (JobResult * n)()
    keep = []
    for i, it in enumerate(items):
        data = bytes(it["input"])
        prefix = bytes(it.get("prefix", b""))
        existing = bytes(it.get("existing", b""))
        limit = it.get("limit")
        if limit is None:
            limit = (1 << 63) - 1
        cap = it.get("out_cap")
        if cap is None:
            cap = len(existing) + min(limit, 1 << 26) + len(data) + 64
        ob = C.create_string_buffer(max(cap, 1))
        ob[: len(existing)] = existing
        keep.append((data, prefix, ob))          # (the bytes objects themselves are the inputs: no staging copy on this side)
        jobs[i].input = _bytes_address(data)
        jobs[i].input_len = len(data)
        jobs[i].prefix = _bytes_address(prefix)
        jobs[i].prefix_len = len(prefix)
        jobs[i].out = C.addressof(ob)
        jobs[i].out_existing_len = len(existing)
        jobs[i].out_cap = cap
        jobs[i].output_limit = limit
    check(lib().lzf_decompress_batch_host(jobs, res, n))
    out = []
    for i in range(n):
        ln = min(res[i].out_len, jobs[i].out_cap)
        out.append((res[i].status, C.string_at(keep[i][2], ln)))
        keep[i] = None                           # a block's staging buffer goes as soon as its bytes are out (streaming readers: memory = blocks in flight)
    return out
