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


class LZ4FrameWriter:
    """lzf_frame_writer_*: compress_internal's loop (src/framed/compress.rs:160-282) with the stream fed piece by piece."""
    _WRITE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)

    def __init__(self, settings, sink, content_size=None, blocks_per_launch=0):
        self._exc = None

        def cb(ctx, p, n):
            try:
                sink(C.string_at(p, n))         # (one memcpy; a slice of the POINTER would build a list of n Python ints first)
                return 0
            except BaseException as e:          # the sink refuses: the writer stops (io::Error of the reference's writer);
                self._exc = e                   # KeyboardInterrupt and friends are re-raised by _done as well
                return 1
        self._cb = self._WRITE(cb)
        self._s = settings._struct(content_size)
        self._w = C.c_void_p()
        L = ffi.lib()
        L.lzf_frame_writer_new.argtypes = [C.c_void_p, self._WRITE, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.lzf_frame_writer_write.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.lzf_frame_writer_finish.argtypes = [C.c_void_p]
        L.lzf_frame_writer_free.argtypes = [C.c_void_p]
        rc = L.lzf_frame_writer_new(C.byref(self._s), self._cb, None, blocks_per_launch, C.byref(self._w))
        if rc != 0:
            ffi.check(rc)
            raise FrameError(rc)

    def _done(self, rc):
        if self._exc is not None:               # whatever the sink raised comes back to the caller, whatever status the C side made of it
            e, self._exc = self._exc, None
            raise e
        if rc != 0:
            ffi.check(rc)
            raise FrameError(rc)

    def write(self, data):
        data = bytes(data)
        self._done(ffi.lib().lzf_frame_writer_write(self._w, data, len(data)))

    def finish(self):
        self._done(ffi.lib().lzf_frame_writer_finish(self._w))

    def close(self):
        if self._w:
            ffi.lib().lzf_frame_writer_free(self._w)
            self._w = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
            s.dictionary = C.addressof(keep)
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
        return C.string_at(out, n.value)

    def compress(self, data):                    # :137-140
        return self._run(data, None)

    def writer(self, sink, content_size=None, blocks_per_launch=0):
        """Streaming form of compress / compress_with_size_unchecked (:138-146): feed the stream with .write(bytes) in any
        granularity, call .finish(); the frame's bytes go to `sink(bytes)` (raise to refuse) piece by piece, in the
        reference's order (lzf_frame_writer_* of lzfear_frame.h).  Byte-identical to compress() of the whole stream."""
        return LZ4FrameWriter(self, sink, content_size, blocks_per_launch)

    def compress_many(self, datas):
        """`compress` of every buffer in `datas`, all frames through the same launches (lzf_frame_compress_many): the
        way to keep the device busy when single frames have only a few blocks, and the only way for linked-block
        streams (block k of every stream goes into launch k).  Returns one frame per input, byte-identical to
        `compress(d)` of each."""
        datas = [bytes(d) for d in datas]
        n = len(datas)
        if n == 0:
            return []
        s = self._struct(None)
        L = ffi.lib()
        caps = [L.lzf_frame_compress_bound(C.byref(s), len(d)) for d in datas]
        outs = [C.create_string_buffer(max(c, 1)) for c in caps]
        ins = (C.c_char_p * n)(*datas)
        lens = (C.c_size_t * n)(*[len(d) for d in datas])
        outp = (C.c_void_p * n)(*[C.addressof(o) for o in outs])
        capa = (C.c_size_t * n)(*caps)
        olen = (C.c_size_t * n)()
        st = (C.c_int * n)()
        ffi.check(L.lzf_frame_compress_many(C.byref(s), n, ins, lens, outp, capa, olen, st))
        for f in range(n):
            if st[f] != 0:
                raise FrameError(st[f])
        return [C.string_at(outs[f], olen[f]) for f in range(n)]

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
        raise FrameError(rc, C.string_at(out, n.value))
    return C.string_at(out, n.value)


def decompress_frames(frames, dictionary=b"", caps=None, with_consumed=False):
    """`decompress_frame` of every frame through the same launches (lzf_frame_decompress_many).  Returns a list of
    (status, bytes): status 0 and the content, or the inner error kind and what the blocks before it produced —
    exactly what `decompress_frame` returns / raises for each frame alone."""
    frames = [bytes(f) for f in frames]
    dictionary = bytes(dictionary)
    n = len(frames)
    if n == 0:
        return []
    if caps is None:
        caps = [min(max(1 << 20, len(f) * 300 + (8 << 20)), 1 << 30) for f in frames]
    outs = [C.create_string_buffer(c) for c in caps]
    ins = (C.c_char_p * n)(*frames)
    lens = (C.c_size_t * n)(*[len(f) for f in frames])
    outp = (C.c_void_p * n)(*[C.addressof(o) for o in outs])
    capa = (C.c_size_t * n)(*caps)
    olen = (C.c_size_t * n)()
    used = (C.c_size_t * n)()
    st = (C.c_int * n)()
    ffi.check(ffi.lib().lzf_frame_decompress_many(n, ins, lens, dictionary, len(dictionary), outp, capa, olen, used, st))
    if with_consumed:
        return [(st[f], C.string_at(outs[f], olen[f]), used[f]) for f in range(n)]
    return [(st[f], C.string_at(outs[f], olen[f])) for f in range(n)]


class FrameBlockReader:
    """The C ABI's block-by-block reader (lzf_frame_reader_*, include/lzfear_frame.h) over a frame in memory:
    `LZ4FrameReader::new` + `decode_block` (src/framed/decompress.rs:102-161,198-282), one call = one block."""

    def __init__(self, frame):
        self._frame = bytes(frame)                       # kept alive: the reader points into it
        self._h = C.c_void_p()
        rc = ffi.lib().lzf_frame_reader_new(self._frame, len(self._frame), C.byref(self._h))
        if rc != 0:
            raise FrameError(rc)
        self.info = ffi.FrameInfo()
        ffi.lib().lzf_frame_reader_info(self._h, C.byref(self.info))
        self._buf = C.create_string_buffer(int(self.info.block_maxsize) * 2 + 64)

    def decode_block(self, dictionary=b""):
        """One block (b"" once the frame is finished — or for a block that decodes to nothing); raises FrameError."""
        n = C.c_size_t(0)
        dictionary = bytes(dictionary)
        rc = ffi.lib().lzf_frame_reader_decode_block(self._h, dictionary, len(dictionary), self._buf, len(self._buf), C.byref(n))
        ffi.check(rc)
        if rc != 0:
            raise FrameError(rc)
        return C.string_at(self._buf, n.value)

    def finished(self):
        return bool(ffi.lib().lzf_frame_reader_finished(self._h))

    def consumed(self):
        return int(ffi.lib().lzf_frame_reader_consumed(self._h))

    def close(self):
        if self._h:
            ffi.lib().lzf_frame_reader_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LZ4FrameReader:
    """Block-by-block frame reader — `LZ4FrameReader` + `LZ4FrameIoReader` of the reference
    (src/framed/decompress.rs:46-77,82-280) over a file-like object: `read(n)`, `fill_buf()`,
    `consume(n)`, `block_size()`, `frame_size()`, `dictionary_id()`.

    The reference decodes one block per `fill_buf`; one GPU launch per block would waste the device,
    so independent-block frames are read ahead `readahead` blocks at a time and decoded as one batch
    (errors still surface at the block where the reference would report them).  Linked-block frames
    are sequential by construction and go block by block with the 64 KiB window as prefix."""

    def __init__(self, reader, dictionary=b"", readahead=16):
        self._r = reader
        self._dict = bytes(dictionary)
        self._readahead = max(1, int(readahead))
        hdr = self._read_exact(7)
        flg = hdr[4] if len(hdr) > 4 else 0
        extra = (8 if flg & 0x08 else 0) + (4 if flg & 0x01 else 0)
        if len(hdr) == 7 and extra:
            hdr += self._read_exact(extra, allow_short=True)
        self.info = ffi.FrameInfo()
        rc = ffi.lib().lzf_frame_read_header(hdr, len(hdr), C.byref(self.info))
        if rc != 0:
            raise FrameError(rc)
        self._linked = not (self.info.flags & 0x20)
        self._bsum = bool(self.info.flags & 0x10)
        self._csum = bool(self.info.flags & 0x04)
        self._hasher = ffi.Xxh32State()
        ffi.lib().lzf_xxh32_reset(C.byref(self._hasher), 0)
        self._window = b""                 # carryover_window (:90,:253-269)
        self._ready = []                   # decoded blocks waiting to be handed out
        self._pending_error = None
        self._buffer = b""
        self._taken = 0
        self._finished = False

    # ---- accessors (:167-175)
    def block_size(self):
        return int(self.info.block_maxsize)

    def frame_size(self):
        return int(self.info.content_size) if self.info.has_content_size else None

    def dictionary_id(self):
        return int(self.info.dictionary_id) if self.info.has_dictionary_id else None

    def _read_exact(self, n, allow_short=False):
        out = b""
        while len(out) < n:
            c = self._r.read(n - len(out))
            if not c:
                break
            out += c
        return out

    def _scan_block(self):
        """One block of the wire format (:205-235) -> ('end', checksum|None) | ('blk', data, compressed)."""
        w = self._read_exact(4)
        if len(w) < 4:
            raise FrameError(16)
        bl = int.from_bytes(w, "little")
        if bl == 0:
            want = None
            if self._csum:
                c = self._read_exact(4)
                if len(c) < 4:
                    raise FrameError(16)
                want = int.from_bytes(c, "little")
            return ("end", want)
        compressed = not (bl & 0x80000000)
        bl &= 0x7FFFFFFF
        if bl > self.block_size():
            raise FrameError(22)
        data = self._read_exact(bl)
        if len(data) < bl:
            raise FrameError(16)
        if self._bsum:
            c = self._read_exact(4)
            if len(c) < 4:
                raise FrameError(16)
            return ("blk", data, compressed, int.from_bytes(c, "little"))       # verified by _refill, on the device
        return ("blk", data, compressed, None)

    def _refill(self):
        """Scan up to `readahead` blocks and decode the compressed ones in one batch (one block at a time
        for linked frames).  Results, an EndMark or the first error are queued in stream order."""
        scanned, tail = [], None
        limit = 1 if self._linked else self._readahead
        while len(scanned) < limit:
            try:
                b = self._scan_block()
            except FrameError as e:
                tail = e                      # reported after the blocks before it have been delivered
                break
            if b[0] == "end":
                tail = b
                break
            scanned.append(b)
        if self._bsum and scanned:                                  # :228-235, all scanned blocks in one device launch
            got = ffi.xxh32_blocks_host([b[1] for b in scanned])
            for k, (b, h) in enumerate(zip(scanned, got)):
                if b[3] != h:
                    scanned, tail = scanned[:k], FrameError(19)     # reported after the blocks before it
                    break
        bmax = self.block_size()
        if self._linked and not self._window:
            self._window = self._dict                               # :239-241
        prefix = self._window if self._linked else self._dict       # :238-245
        items = [dict(input=d, prefix=prefix, limit=bmax, out_cap=bmax + len(d)) for (_, d, comp, _c) in scanned if comp]
        res = iter(ffi.decompress_blocks_host(items)) if items else iter(())
        for (_, d, comp, _c) in scanned:
            if comp:
                rc, out = next(res)
                if rc != 0:
                    self._ready.append(FrameError(rc))              # CodecError
                    return
            else:
                out = d                                             # :250
            if self._linked:                                        # window update :253-269
                self._window = (self._window + out)[-WINDOW_SIZE:] if len(out) < WINDOW_SIZE else out[-WINDOW_SIZE:]
            if len(out) > bmax:
                self._ready.append(FrameError(22))                  # :272-274
                return
            self._ready.append(out)
        if tail is not None:
            self._ready.append(tail)

    def fill_buf(self):
        """BufRead::fill_buf (:64-71): the current decoded block (b"" at the end of the frame)."""
        if self._taken == len(self._buffer) and not self._finished:
            self._buffer, self._taken = b"", 0
            if not self._ready:
                self._refill()
            item = self._ready[0]
            if isinstance(item, FrameError):
                raise item                                          # sticky
            if isinstance(item, tuple):                             # EndMark: verify the content checksum (:206-213)
                if item[1] is not None and item[1] != ffi.lib().lzf_xxh32_digest(C.byref(self._hasher)):
                    self._ready[0] = FrameError(20)
                    raise self._ready[0]
                self._ready.pop(0)
                self._finished = True
            else:
                self._ready.pop(0)
                if self._csum:
                    ffi.lib().lzf_xxh32_update(C.byref(self._hasher), item, len(item))    # :276-278
                self._buffer = item
        return self._buffer[self._taken:] if self._taken else self._buffer      # (no copy of a block nobody has consumed from)

    def consume(self, amt):
        self._taken += amt
        assert self._taken <= len(self._buffer), "You consumed more bytes than I even gave you!"   # :75

    def read(self, n=-1):
        """io::Read::read (:54-60) when n >= 0; read_to_end when n < 0 (a 0-byte block ends it, like the
        reference's read_to_end)."""
        if n is None or n < 0:
            out = b""
            while True:
                b = self.fill_buf()
                if not b:
                    return out
                out += b
                self.consume(len(b))
        b = self.fill_buf()
        take = min(len(b), n)
        self.consume(take)
        return b[:take]
