"""Mirror of lz-fear's `framed` module on top of the C ABI (include/lzfear_frame.h):
`CompressionSettings` with the reference's builder methods (src/framed/compress.rs:56-157) and
`decompress_frame` (src/framed/decompress.rs:284-288).  Thin Python over the C++ frame layer —
every block is compressed / decompressed by the HIP kernels."""
import ctypes as C

from . import ffi

MAGIC = 0x184D2204          # src/framed/mod.rs:16
WINDOW_SIZE = 64 * 1024     # src/framed/mod.rs:20

FRAME_ERRORS = {
    1: "CodecError(UnexpectedEnd)", 2: "CodecError(MemoryLimitExceeded)", 3: "CodecError(ZeroDeduplicationOffset)",
    4: "CodecError(InvalidDeduplicationOffset)", 7: "OutCapacity",
    16: "InputError", 17: "WrongMagic", 18: "HeaderChecksumFail", 19: "BlockChecksumFail", 20: "FrameChecksumFail",
    21: "BlockLengthOverflow", 22: "BlockSizeOverflow", 23: "UnimplementedBlocksize", 24: "UnsupportedVersion",
    25: "ReservedFlagBitsSet", 26: "ReservedBdBitsSet", 27: "InvalidBlockSize", 28: "Panic",
}


class FrameError(Exception):
    def __init__(self, code, partial=b""):
        super().__init__(FRAME_ERRORS.get(code, str(code)))
        self.code = code
        self.partial = partial


class CompressionSettings:
    """src/framed/compress.rs:36-55; the setters return self like the reference's builder."""

    def __init__(self):
        self._independent_blocks = True
        self._block_checksums = False
        self._content_checksum = True
        self._block_size = 4 * 1024 * 1024
        self._dictionary = None
        self._dictionary_id = None

    def independent_blocks(self, v):            # :63-66
        self._independent_blocks = bool(v); return self

    def block_checksums(self, v):               # :75-78
        self._block_checksums = bool(v); return self

    def content_checksum(self, v):              # :88-91
        self._content_checksum = bool(v); return self

    def block_size(self, v):                    # :97-100
        self._block_size = int(v); return self

    def dictionary(self, id, dict):             # :113-117
        self._dictionary_id = id; self._dictionary = bytes(dict); return self

    def dictionary_id_nonsense_override(self, id):   # :130-133
        self._dictionary_id = id; return self

    def _struct(self, content_size):
        s = ffi.Settings()
        ffi.lib().lzf_settings_default(C.byref(s))
        s.independent_blocks = int(self._independent_blocks)
        s.block_checksums = int(self._block_checksums)
        s.content_checksum = int(self._content_checksum)
        s.block_size = self._block_size
        keep = None
        if self._dictionary is not None:
            keep = C.create_string_buffer(self._dictionary, max(len(self._dictionary), 1))
            s.dictionary = C.cast(keep, C.c_void_p)
            s.dictionary_len = len(self._dictionary)
        if self._dictionary_id is not None:
            s.has_dictionary_id = 1
            s.dictionary_id = self._dictionary_id
        if content_size is not None:
            s.has_content_size = 1
            s.content_size = content_size
        s._keep = keep
        return s

    def _run(self, data, content_size):
        data = bytes(data)
        s = self._struct(content_size)
        cap = ffi.lib().lzf_frame_compress_bound(C.byref(s), len(data))
        out = C.create_string_buffer(max(cap, 1))
        n = C.c_size_t(0)
        rc = ffi.lib().lzf_frame_compress(C.byref(s), data, len(data), out, cap, C.byref(n))
        ffi.check(rc)
        if rc != 0:
            raise FrameError(rc)
        return out.raw[: n.value]

    def compress(self, data):                    # :137-140
        return self._run(data, None)

    def compress_with_size(self, data):          # :147-157
        return self._run(data, len(data))

    def compress_with_size_unchecked(self, data, content_size):   # :142-145
        return self._run(data, content_size)


def read_header(frame):
    """LZ4FrameReader::new (src/framed/decompress.rs:102-161) -> ffi.FrameInfo."""
    frame = bytes(frame)
    info = ffi.FrameInfo()
    rc = ffi.lib().lzf_frame_read_header(frame, len(frame), C.byref(info))
    if rc != 0:
        raise FrameError(rc)
    return info


def decompress_frame(frame, dictionary=b"", cap=None):
    """decompress_frame (src/framed/decompress.rs:284-288); raises FrameError with the inner kind."""
    frame = bytes(frame)
    dictionary = bytes(dictionary)
    if cap is None:
        cap = min(max(1 << 20, len(frame) * 300 + (8 << 20)), 1 << 30)
    out = C.create_string_buffer(cap)
    n = C.c_size_t(0)
    used = C.c_size_t(0)
    rc = ffi.lib().lzf_frame_decompress(frame, len(frame), dictionary, len(dictionary), out, cap, C.byref(n), C.byref(used))
    ffi.check(rc)
    if rc != 0:
        raise FrameError(rc, out.raw[: n.value])
    return out.raw[: n.value]
