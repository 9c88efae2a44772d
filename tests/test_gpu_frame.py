"""GPU tests (-m gpu) of the frame layer (include/lzfear_frame.h) — lz-fear's `framed` module with
every block coded by the HIP kernels — against the oracle's frame restatement: exact frame bytes
over the flag matrix of tests/output_equivalence.rs, the issue-15 regression, the survey
fingerprints, dictionary modes, and malformed frames with pinned error kinds."""
import json
import os

import numpy as np
import pytest

import oracle_ffi as o
import vectors
import rust_lz_fear_amd  # noqa: F401
from rust_lz_fear_amd import ffi, framed, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
fp = lambda b: [len(b), "%08x" % o.xxh32(b)]


def settings_pair(**kw):
    """(GPU-side CompressionSettings, oracle settings) from the same keyword set."""
    g = framed.CompressionSettings()
    if "independent_blocks" in kw: g.independent_blocks(kw["independent_blocks"])
    if "block_checksums" in kw: g.block_checksums(kw["block_checksums"])
    if "content_checksum" in kw: g.content_checksum(kw["content_checksum"])
    if "block_size" in kw: g.block_size(kw["block_size"])
    if "dictionary" in kw:
        g.dictionary(kw.get("dictionary_id", 0), kw["dictionary"])
        if kw.get("dictionary_id") is None:
            g.dictionary_id_nonsense_override(None)
    okw = {k: v for k, v in kw.items()}
    return g, okw


@pytest.mark.parametrize("bits", range(32))
def test_flag_matrix_exact_frame_bytes(bits):
    """tests/output_equivalence.rs:58-101 flag matrix (bit 2 = block size, run here at 64/256 KiB)."""
    data = synth.silesia_mix(0, 700_000).tobytes()
    kw = dict(content_checksum=not (bits & 1), independent_blocks=not (bits & 2),
              block_size=(256 << 10) if bits & 4 else (64 << 10))
    if bits & 8:
        kw["dictionary"] = bytes([1, 3, 3, 7]); kw["dictionary_id"] = None
    g, okw = settings_pair(**kw)
    size = len(data) if bits & 16 else None
    got = g.compress_with_size(data) if size is not None else g.compress(data)
    rc, want = o.frame_compress(data, o.make_settings(content_size=size, **okw))
    assert rc == 0 and got == want
    assert framed.decompress_frame(got, dictionary=kw.get("dictionary", b"")) == data


@pytest.mark.parametrize("bits", range(32))
def test_streaming_writer_flag_matrix_exact_frame_bytes(bits):
    """lzf_frame_writer_* (the streaming form of CompressionSettings::compress, compress.rs:138-157,:221-276): the same flag
    matrix, the stream fed in three granularities (7 bytes to whole blocks and more), launches of 1, 3 and 64 blocks — every
    time the oracle's frame, byte for byte, and the one-call lzf_frame_compress's."""
    data = synth.silesia_mix(0, 700_000).tobytes()
    kw = dict(content_checksum=not (bits & 1), independent_blocks=not (bits & 2),
              block_size=(256 << 10) if bits & 4 else (64 << 10), block_checksums=bool(bits & 1))
    if bits & 8:
        kw["dictionary"] = bytes([1, 3, 3, 7]) if bits & 4 else synth.silesia_mix(900_000, 930_000).tobytes(); kw["dictionary_id"] = None
    g, okw = settings_pair(**kw)
    size = len(data) if bits & 16 else None
    rc, want = o.frame_compress(data, o.make_settings(content_size=size, **okw))
    assert rc == 0
    assert (g.compress_with_size(data) if size is not None else g.compress(data)) == want
    rng = np.random.default_rng(bits)
    for per_launch, feed in ((1, "tiny"), (3, "ragged"), (0, "big")):
        got = bytearray()
        w = g.writer(got.extend, content_size=size, blocks_per_launch=per_launch)
        pos = 0
        while pos < len(data):
            n = 7 if (feed == "tiny" and pos < 300) else int(rng.integers(1, 90_000)) if feed != "big" else 300_001
            w.write(data[pos:pos + n]); pos += n
        w.finish(); w.close()
        assert bytes(got) == want, (bits, feed)
    # the empty stream: header, EndMark, checksum
    got = bytearray(); w = g.writer(got.extend, content_size=0 if size is not None else None); w.finish(); w.close()
    assert bytes(got) == o.frame_compress(b"", o.make_settings(content_size=0 if size is not None else None, **okw))[1]


def test_streaming_writer_refusing_sink_stops_the_writer():
    """A sink that refuses (raises) after some bytes: the write that crosses it reports the sink's error, what was accepted
    is a prefix of the frame, and the writer is dead."""
    data = synth.silesia_mix(0, 300_000).tobytes()
    g, okw = settings_pair(block_size=64 << 10)
    want = o.frame_compress(data, o.make_settings(**okw))[1]
    for budget in (0, 5, 7, 11, 40_000, len(want) - 3):
        got = bytearray()
        def sink(b):
            if len(got) + len(b) > budget:
                raise IOError("full")
            got.extend(b)
        w = g.writer(sink, blocks_per_launch=2)
        with pytest.raises(IOError):
            w.write(data); w.finish()
        assert want.startswith(bytes(got)) and len(got) <= budget
        with pytest.raises(Exception):
            w.write(b"x")
        w.close()


def test_default_settings_4mib_blocks():
    data = synth.silesia_mix(5 << 20, 15 << 20).tobytes()          # 2.5 blocks of 4 MiB
    got = framed.CompressionSettings().compress(data)
    rc, want = o.frame_compress(data)
    assert rc == 0 and got == want
    assert framed.decompress_frame(got) == data


def test_survey_frame_fingerprints():
    S = json.load(open(os.path.join(GOLD, "survey_fingerprints.json")))
    g3 = synth.lcg_bytes(5, 262144, 3)
    assert fp(framed.CompressionSettings().block_size(65536).compress(g3)) == S["G3"]["frame_64k_independent"]
    f = framed.CompressionSettings().block_size(65536).independent_blocks(False).block_checksums(True).compress_with_size(g3)
    assert fp(f) == S["G3"]["frame_64k_linked_blocksum_csize"]
    assert framed.decompress_frame(f) == g3
    kb = synth.lcg_bytes(3, 69632, 3)
    f = framed.CompressionSettings().block_size(65536).independent_blocks(False).compress(kb)
    assert fp(f) == S["KAT-B"]["frame_64k_linked"]                  # quirks B1/B3 through the GPU table carry
    assert framed.decompress_frame(f) == kb


def test_issue15_regression_linked_64k():
    data = open(os.path.join(GOLD, "issue15_input.bin"), "rb").read()
    f = framed.CompressionSettings().independent_blocks(False).block_size(64 * 1024).compress(data)   # tests/issue-15.rs:10-13
    assert len(f) == 81160
    assert f == o.frame_compress(data, o.make_settings(independent_blocks=False, block_size=64 * 1024))[1]
    assert framed.decompress_frame(f) == data                       # :15-21


def test_corpus_frame_and_checksum_failure():
    data = open(os.path.join(GOLD, "uncomp.data.lz4"), "rb").read()
    J = json.load(open(os.path.join(GOLD, "corpus_frames.json")))["valid_frames"]["uncomp.data.lz4"]
    assert fp(framed.decompress_frame(data)) == [J["out_len"], J["out_xxh32"]]
    bad = bytearray(data); bad[20] ^= 1
    with pytest.raises(framed.FrameError) as e:
        framed.decompress_frame(bytes(bad))
    assert e.value.code == o.F_FRAME_CHECKSUM_FAIL
    info = framed.read_header(data)
    assert info.block_maxsize == 64 << 10 and info.flags == 0x64 and info.bd == 0x40


def test_dictionary_modes():
    d = synth.gen_text_zipf(3, 70000).tobytes()
    data = synth.gen_text_zipf(4, 300000).tobytes()
    for indep in (True, False):
        g = framed.CompressionSettings().independent_blocks(indep).block_size(64 << 10).dictionary(42, d)
        f = g.compress(data)
        rc, want = o.frame_compress(data, o.make_settings(independent_blocks=indep, block_size=64 << 10, dictionary=d, dictionary_id=42))
        assert rc == 0 and f == want
        assert framed.decompress_frame(f, dictionary=d) == data
        assert framed.read_header(f).dictionary_id == 42
        with pytest.raises(framed.FrameError):
            framed.decompress_frame(f)                               # missing dictionary


def test_block_size_validation():
    for bs in (1, 1000, 32 << 10, 2 << 20, 8 << 20):
        with pytest.raises(framed.FrameError) as e:
            framed.CompressionSettings().block_size(bs).compress(b"x")
        assert e.value.code == o.F_INVALID_BLOCK_SIZE
    for bs in (0, 16 << 20):
        with pytest.raises(framed.FrameError) as e:
            framed.CompressionSettings().block_size(bs).compress(b"x")
        assert e.value.code == o.F_PANIC


def test_incompressible_and_empty():
    data = vectors.rng_bytes(5, 200000)
    f = framed.CompressionSettings().block_size(64 << 10).compress(data)
    assert f == o.frame_compress(data, o.make_settings(block_size=64 << 10))[1]
    assert framed.decompress_frame(f) == data
    e = framed.CompressionSettings().compress(b"")
    assert e == o.frame_compress(b"")[1] and framed.decompress_frame(e) == b""


def mutate(rng, frame):
    b = bytearray(frame)
    kind = rng.integers(0, 5)
    if kind == 0 and len(b) > 8:
        del b[rng.integers(7, len(b)):]
    elif kind == 1:
        i = rng.integers(0, len(b)); b[i] ^= 1 << rng.integers(0, 8)
    elif kind == 2:
        i = rng.integers(4, min(len(b), 12)); b[i] = rng.integers(0, 256)       # header bytes
    elif kind == 3:
        i = rng.integers(0, len(b)); b[i] = 0 if rng.integers(0, 2) else 0xFF
    else:
        for _ in range(3):
            i = rng.integers(0, len(b)); b[i] = rng.integers(0, 256)
    return bytes(b)


def test_malformed_frames_same_error_kind_and_partial_output():
    """fuzz/fuzz_targets/decode.rs idea with pinned kinds: every mutated frame must fail (or
    succeed) exactly like the reference restatement, block by block."""
    rng = np.random.default_rng(777)
    base = []
    data = synth.silesia_mix(40 << 20, (40 << 20) + 300_000).tobytes()
    for kw in (dict(block_size=64 << 10), dict(block_size=64 << 10, independent_blocks=False),
               dict(block_size=64 << 10, block_checksums=True), dict(block_size=256 << 10, content_checksum=False)):
        base.append(o.frame_compress(data, o.make_settings(**kw))[1])
    base.append(o.frame_compress(vectors.rng_bytes(2, 100000), o.make_settings(block_size=64 << 10))[1])
    kinds = set()
    for f in base:
        for _ in range(40):
            m = mutate(rng, f)
            erc, eout, _ = o.frame_decompress(m, cap=16 << 20)
            try:
                out = framed.decompress_frame(m, cap=16 << 20); rc = 0
            except framed.FrameError as e:
                rc, out = e.code, e.partial
            assert rc == erc, (rc, erc, len(m))
            assert out == eout
            kinds.add(rc)
    assert len(kinds) >= 8, kinds


# ---------------------------------------------------------------- BASELINE.json configs as parity cases
def test_config1_text_64k_default_frame():
    """configs[0]: one 64 KiB block of enwik8-like text, default CompressionSettings."""
    data = synth.text_zipf_64k().tobytes()
    f = framed.CompressionSettings().compress(data)
    assert f == o.frame_compress(data)[1]
    assert framed.read_header(f).block_maxsize == 4 << 20
    assert framed.decompress_frame(f) == data


def test_config4_log_text_blocks_and_sharded_assembly():
    """configs[3] in miniature: log-text stream, 4 MiB independent blocks; the frame assembled from
    block-sharded GPU output (what the RCCL all-gather reassembles) equals the single-call frame."""
    import ctypes as C
    from rust_lz_fear_amd import ffi
    bs = 4 << 20
    data = synth.log_text(3 * bs, 3 * bs + 2 * bs + 777_777).tobytes()
    whole = framed.CompressionSettings().compress(data)
    assert whole == o.frame_compress(data)[1]
    nblk = (len(data) + bs - 1) // bs
    blocks = [data[i * bs:(i + 1) * bs] for i in range(nblk)]
    res = ffi.compress_blocks_host([dict(input=b, out_cap=len(b)) for b in blocks])      # "rank-local" batches
    payload = [c if rc == 0 else b for (rc, c), b in zip(res, blocks)]
    bufs = [C.create_string_buffer(p, max(len(p), 1)) for p in payload]
    ptrs = (C.c_void_p * nblk)(*[C.cast(b, C.c_void_p) for b in bufs])
    cl = (C.c_uint32 * nblk)(*[len(c) if rc == 0 else 0xFFFFFFFF for rc, c in res])
    rl = (C.c_uint32 * nblk)(*[len(b) for b in blocks])
    s = framed.CompressionSettings()._struct(None)
    out = C.create_string_buffer(len(data) + 1024)
    n = C.c_size_t(0)
    assert ffi.lib().lzf_frame_assemble(C.byref(s), nblk, ptrs, cl, rl, ffi.lib().lzf_xxh32(data, len(data), 0), out, len(out), C.byref(n)) == 0
    assert out.raw[: n.value] == whole
    assert framed.decompress_frame(whole) == data


def test_config5_repeat256_linked_dictionary_and_u16():
    """configs[4]: 256-byte motif; (5a) framed 64 KiB linked blocks with a 64 KiB dictionary of the
    same motif (the U32Table linked path, the only one the reference's frame layer has);
    (5b) raw compress2::<U16Table> on 65 535-byte slices, cursor 0 and behind a 4096-byte prefix."""
    from rust_lz_fear_amd import ffi
    motif_dict = synth.repeat256(65536).tobytes()
    data = synth.repeat256(5 * 65536 + 1234).tobytes()
    g = framed.CompressionSettings().independent_blocks(False).block_size(64 << 10).dictionary(7, motif_dict)
    f = g.compress(data)
    rc, want = o.frame_compress(data, o.make_settings(independent_blocks=False, block_size=64 << 10, dictionary=motif_dict, dictionary_id=7))
    assert rc == 0 and f == want and len(f) < 3000
    assert framed.decompress_frame(f, dictionary=motif_dict) == data
    sl = synth.repeat256(65535).tobytes()
    res = ffi.compress_blocks_host([dict(input=sl, kind=ffi.TABLE_U16), dict(input=sl, cursor=4096, kind=ffi.TABLE_U16)])
    assert res[0] == o.compress2(sl, kind=o.TABLE_U16)
    assert res[1] == o.compress2(sl, cursor=4096, kind=o.TABLE_U16)
    back = ffi.decompress_blocks_host([dict(input=res[0][1], limit=65535), dict(input=res[1][1], prefix=sl[:4096], limit=65535 - 4096)])
    assert back[0] == (0, sl) and back[1] == (0, sl[4096:])


def test_streaming_reader_matches_decompress_frame():
    """LZ4FrameReader (block-by-block, read-ahead batches) == decompress_frame, including errors."""
    import io
    data = synth.silesia_mix(60 << 20, (60 << 20) + 1_300_000).tobytes()
    for kw in (dict(block_size=64 << 10), dict(block_size=64 << 10, independent_blocks=False),
               dict(block_size=256 << 10, block_checksums=True), dict(block_size=64 << 10, content_checksum=False)):
        f = o.frame_compress(data, o.make_settings(**kw))[1]
        for ra in (1, 3, 64):
            r = framed.LZ4FrameReader(io.BytesIO(f), readahead=ra)
            assert r.block_size() == kw["block_size"] and r.frame_size() is None and r.dictionary_id() is None
            got = b""
            while True:                                   # examples/delz4.rs loop: fill_buf / consume
                b = r.fill_buf()
                if not b:
                    break
                got += b[:1000]; r.consume(min(1000, len(b)))
            assert got == data
        assert framed.LZ4FrameReader(io.BytesIO(f)).read() == data
        # small reads through io::Read::read
        r = framed.LZ4FrameReader(io.BytesIO(f), readahead=4)
        got = b""
        while True:
            c = r.read(4096)
            if not c:
                break
            got += c
        assert got == data
    # errors surface at the same block with the same kind
    rng = np.random.default_rng(31)
    f = o.frame_compress(data, o.make_settings(block_size=64 << 10, block_checksums=True))[1]
    for _ in range(25):
        m = mutate(rng, f)
        erc, eout, _ = o.frame_decompress(m, cap=16 << 20)
        got, rc = b"", 0
        try:
            r = framed.LZ4FrameReader(io.BytesIO(m), readahead=5)
            got = r.read()
        except framed.FrameError as e:
            rc = e.code
        assert rc == erc, (rc, erc)
        if rc == 0:
            assert got == eout
    d = synth.gen_text_zipf(3, 70000).tobytes()
    f = o.frame_compress(data, o.make_settings(block_size=64 << 10, independent_blocks=False, dictionary=d, dictionary_id=9))[1]
    r = framed.LZ4FrameReader(io.BytesIO(f), dictionary=d)
    assert r.dictionary_id() == 9 and r.read() == data


def test_config4_sharded_frame_over_rccl():
    """BASELINE config 4 on one GPU: blocks of a log-text stream compressed on the device, the size table and
    the payloads all-gathered over torch.distributed's nccl backend (= RCCL; world_size 1 here, the 2-rank
    exchange is covered on gloo by tests/test_dist_gloo.py), the frame assembled — byte-identical to the oracle's
    frame, and it decodes back."""
    import socket
    import torch
    import torch.distributed as dist
    from rust_lz_fear_amd import device, ffi, dist as lzdist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        BS = 4 << 20
        data = synth.log_text(0, 5 * BS + 1_234_567)
        d_in = torch.from_numpy(data).cuda()
        blocks = device.BlockSet(d_in, BS); n = blocks.n
        d_out = torch.empty(n * BS, dtype=torch.uint8, device="cuda")
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
        device.compress_batch(device.to_device(blocks.compress_jobs(d_out, BS), "cuda"), d_res, n)
        torch.cuda.synchronize()
        res = device.results_to_host(d_res, n)
        lo, hi = lzdist.shard_range(n, 0, 1)
        payloads, clens = [], []
        for b in range(lo, hi):
            if res["status"][b] == 0:
                payloads.append(d_out[b * BS: b * BS + int(res["out_len"][b])]); clens.append(int(res["out_len"][b]))
            else:
                payloads.append(d_in[b * BS: min((b + 1) * BS, len(data))]); clens.append(lzdist.STORED)
        allp, allc = lzdist.allgather_blocks(payloads, clens, n, device="cuda")
        raw_len = [min(BS, len(data) - b * BS) for b in range(n)]
        raw = data.tobytes()
        st = framed.CompressionSettings().block_size(BS)._struct(None)
        frame = lzdist.assemble_frame(st, allp, allc, raw_len, ffi.lib().lzf_xxh32(raw, len(raw), 0))
        assert frame == o.frame_compress(raw, o.make_settings(block_size=BS))[1]
        assert framed.decompress_frame(frame) == raw
    finally:
        dist.destroy_process_group()


def test_gather_frame_device_nccl_world1():
    """The function `bench.py --workload config4` times (dist.gather_frame_device: size-table all-gather, in-place packing
    with lzf_copy_ranges, exact-size P2P segments) on the nccl backend at world size 1: the WHOLE frame equals the oracle's
    frame — a stored (incompressible) block and a short last block included — and decodes back.  (World size 2: gloo,
    tests/test_dist_gloo.py.)"""
    import socket
    import torch
    import torch.distributed as dist
    from rust_lz_fear_amd import device, ffi, dist as lzdist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        BS = 4 << 20
        data = np.concatenate([synth.log_text(0, 2 * BS), np.frombuffer(vectors.rng_bytes(5, BS), np.uint8),
                               synth.log_text(2 * BS, 3 * BS + 777_777)])
        d_in = torch.from_numpy(data).cuda()
        blocks = device.BlockSet(d_in, BS); n = blocks.n
        assert n == 5 and int(blocks.lens[-1]) == 777_777
        d_out = torch.empty(n * BS, dtype=torch.uint8, device="cuda")
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
        device.compress_batch(device.to_device(blocks.compress_jobs(d_out, BS), "cuda"), d_res, n)
        torch.cuda.synchronize()
        res = device.results_to_host(d_res, n)
        assert ffi.OUTPUT_FULL in set(int(x) for x in res["status"])          # the random block is stored
        frame = torch.zeros(64 + n * (BS + 8), dtype=torch.uint8, device="cuda")
        header = lzdist.frame_header(content_checksum=False, block_size=BS)
        raw_lens = torch.from_numpy(blocks.lens.astype(np.int64)).cuda()
        flen, ctot = lzdist.gather_frame_device(d_res, d_out, d_in, BS, n, n, frame, dist, 0, 1, device, header, raw_lens=raw_lens)
        torch.cuda.synchronize()
        mine = frame[:flen].cpu().numpy().tobytes()
        raw = data.tobytes()
        rc, ref = o.frame_compress(raw, o.make_settings(block_size=BS, content_checksum=False))
        assert rc == 0 and mine == ref
        assert framed.decompress_frame(mine) == raw
    finally:
        dist.destroy_process_group()


def test_frame_gather_c_abi_rccl_world1():
    """The same exchange under the C ABI (include/lzfear_dist.h, liblzfear_dist.so): lzf_dist_unique_id / lzf_dist_comm_init make an RCCL
    communicator of one rank (no torch.distributed involved), lzf_frame_gather runs the size-table all-gather path, the in-place
    packing kernel + lzf_copy_ranges, header and EndMark: the WHOLE frame equals the oracle's — a stored block and a short last
    block included — ncclCommCount says 1, and a status that cannot be framed is refused."""
    import torch
    from rust_lz_fear_amd import device, ffi, dist as lzdist
    BS = 4 << 20
    data = np.concatenate([synth.log_text(0, 2 * BS), np.frombuffer(vectors.rng_bytes(5, BS), np.uint8),
                           synth.log_text(2 * BS, 3 * BS + 777_777)])
    d_in = torch.from_numpy(data).cuda()
    blocks = device.BlockSet(d_in, BS); n = blocks.n
    d_out = torch.empty(n * BS, dtype=torch.uint8, device="cuda")
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
    device.compress_batch(device.to_device(blocks.compress_jobs(d_out, BS), "cuda"), d_res, n)
    torch.cuda.synchronize()
    res = device.results_to_host(d_res, n)
    assert ffi.OUTPUT_FULL in set(int(x) for x in res["status"])              # the random block is stored
    comm = lzdist.DistComm(None, 0, 1, "cuda")
    try:
        assert comm.count() == 1
        frame = torch.zeros(64 + n * (BS + 8), dtype=torch.uint8, device="cuda")
        header = lzdist.frame_header(content_checksum=False, block_size=BS)
        flen, ctot = lzdist.gather_frame_device_c(comm, d_res, d_out, d_in, BS, n, n, frame, header, last_block_len=int(blocks.lens[-1]))
        mine = frame[:flen].cpu().numpy().tobytes()
        raw = data.tobytes()
        rc, ref = o.frame_compress(raw, o.make_settings(block_size=BS, content_checksum=False))
        assert rc == 0 and mine == ref
        assert ctot == sum(int(l) if st == 0 else int(rl) for l, st, rl in zip(res["out_len"], res["status"], blocks.lens))
        assert framed.decompress_frame(mine) == raw
        # a frame buffer that is too small, and a status that is neither OK nor OUTPUT_FULL
        with pytest.raises(ffi.LzfError):
            lzdist.gather_frame_device_c(comm, d_res, d_out, d_in, BS, n, n, frame[:1000], header, last_block_len=int(blocks.lens[-1]))
        bad = res.copy(); bad["status"][1] = ffi.CONTRACT
        d_bad = device.to_device(bad, "cuda")
        with pytest.raises(ffi.LzfError) as ei:
            lzdist.gather_frame_device_c(comm, d_bad, d_out, d_in, BS, n, n, frame, header, last_block_len=int(blocks.lens[-1]))
        assert ei.value.code == ffi.CONTRACT
        # a header that announces a content checksum (or block checksums): the frame this call writes carries neither
        for kw in (dict(content_checksum=True), dict(block_checksums=True)):
            with pytest.raises(ffi.LzfError) as ei:
                lzdist.gather_frame_device_c(comm, d_res, d_out, d_in, BS, n, n, frame, lzdist.frame_header(block_size=BS, **kw), last_block_len=int(blocks.lens[-1]))
            assert ei.value.code == ffi.E_INVALID
        # ... and the library says which librccl it is bound to
        assert "rccl" in lzdist.rccl_paths()["liblzfear_dist_binds"]
    finally:
        comm.close()


def test_frame_gather_c_abi_rccl_world2():
    """lzf_frame_gather with two ranks over RCCL (tests/dist_world2_check.py: one process per GPU): 7 and 8 blocks, a stored block, a
    short last block, fewer blocks than ranks (a rank with no block at all), 2 W + 1 blocks — the whole frame == the oracle's on every
    rank — and failures that only ONE rank can see (a block status that cannot be framed, a frame buffer that is too small, a header
    that announces a checksum): every rank returns the same code, none is left in the exchange, the communicator works afterwards.
    Needs two GPUs: skipped on the one-GPU box, runs on the driver's multi-GPU node."""
    import socket
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs at least two GPUs (one process per GPU over RCCL)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    world = 2
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "dist_world2_check.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank of the two-rank exchange did not return (a hang is exactly what this test is for)")
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{out[-4000:]}"
    assert "world2 ok" in outs[0]


# ---------------------------------------------------------------- many frames per call
def _frame_inputs():
    mix = synth.silesia_mix(20 << 20, (20 << 20) + 900_000).tobytes()
    return [mix[:300_000], b"", mix[300_000:300_017], vectors.rng_bytes(9, 150_000), mix[100_000:760_001],
            synth.repeat256(5 * 65536 + 1234).tobytes(), bytes(200_000), mix[:65536], mix[5:70_000]]


@pytest.mark.parametrize("kw", [dict(block_size=64 << 10), dict(block_size=64 << 10, independent_blocks=False),
                                dict(block_size=256 << 10, independent_blocks=False, block_checksums=True),
                                dict(block_size=64 << 10, dictionary=synth.repeat256(65536).tobytes(), dictionary_id=7),
                                dict(block_size=64 << 10, independent_blocks=False, dictionary=synth.silesia_mix(0, 100_000).tobytes(), dictionary_id=3),
                                dict(block_size=64 << 10, independent_blocks=False, dictionary=b"abcd", dictionary_id=1),
                                dict()])
def test_compress_many_equals_one_frame_at_a_time(kw):
    """lzf_frame_compress_many: every frame byte-identical to the oracle's frame (and so to lzf_frame_compress), in
    independent mode (one launch for all blocks) and linked mode (block k of every stream in launch k, tables and
    windows on the device; streams of different lengths, empty ones, stored blocks, dictionaries)."""
    g, okw = settings_pair(**kw)
    datas = _frame_inputs()
    frames = g.compress_many(datas)
    assert len(frames) == len(datas)
    for d, f in zip(datas, frames):
        assert f == o.frame_compress(d, o.make_settings(**okw))[1], (len(d), kw.keys())


def test_many_frames_without_any_worker_thread():
    """"no hidden host threads required".  The staging's worker threads are a throughput option:
    lzf_frame_set_host_threads(LZF_HOST_THREADS_NONE) makes the calling thread do every pageable <-> pinned copy (and hash)
    itself; frames large enough for the piecewise staging path must come out byte-identical, both ways."""
    L = ffi.lib()
    mix = synth.silesia_mix(3 << 20, 28 << 20).tobytes()
    datas = [mix[: 9 << 20], mix[9 << 20: 20 << 20], mix[20 << 20: 25 << 20], b"", mix[:70_000]]
    g, okw = settings_pair(block_size=1 << 20)
    try:
        L.lzf_frame_set_host_threads(0xFFFFFFFF)
        frames = g.compress_many(datas)
        got = framed.decompress_frames(frames, caps=[len(d) + 64 for d in datas])
    finally:
        L.lzf_frame_set_host_threads(0)
    for d, f, (rc, out) in zip(datas, frames, got):
        assert f == o.frame_compress(d, o.make_settings(**okw))[1], len(d)
        assert rc == 0 and out == d, len(d)
    again = g.compress_many(datas)                      # back on the default pool: the same frames
    assert again == frames


def test_decompress_many_equals_one_frame_at_a_time():
    """lzf_frame_decompress_many over good and damaged frames of every flavour (independent / linked, stored blocks,
    block checksums): the same (status, bytes) as the oracle's decoder reports for each frame alone."""
    rng = np.random.default_rng(4242)
    datas = _frame_inputs()
    frames = []
    for kw in (dict(block_size=64 << 10), dict(block_size=64 << 10, independent_blocks=False),
               dict(block_size=64 << 10, independent_blocks=False, block_checksums=True), dict(block_size=256 << 10, content_checksum=False)):
        for d in datas:
            frames.append(o.frame_compress(d, o.make_settings(**kw))[1])
    damaged = [mutate(rng, f) for f in frames for _ in range(3)]
    allf = frames + damaged
    got = framed.decompress_frames(allf, caps=[4 << 20] * len(allf))
    kinds = set()
    for f, (rc, out) in zip(allf, got):
        erc, eout, _ = o.frame_decompress(f, cap=4 << 20)
        assert rc == erc, (rc, erc, len(f))
        assert out == eout
        kinds.add(rc)
    assert len(kinds) >= 6, kinds


def test_many_frames_with_dictionary_roundtrip_linked():
    """configs[4] shape, many streams in flight: 24 linked 64 KiB-block streams with the motif dictionary."""
    d = synth.repeat256(65536).tobytes()
    g, okw = settings_pair(block_size=64 << 10, independent_blocks=False, dictionary=d, dictionary_id=5)
    datas = [synth.repeat256(65536 * (1 + i % 5) + 17 * i).tobytes()[i:] for i in range(24)]
    frames = g.compress_many(datas)
    for x, f in zip(datas, frames):
        assert f == o.frame_compress(x, o.make_settings(**okw))[1]
    got = framed.decompress_frames(frames, dictionary=d, caps=[1 << 20] * len(frames))
    assert [rc for rc, _ in got] == [0] * len(frames)
    assert [out for _, out in got] == datas


# ---------------------------------------------------------------- round 2: device checksums, staging, reader, budgets
def test_block_and_content_checksums_are_computed_on_the_device():
    """The frame drivers hash every block of a call with lzf_xxh32_batch (one launch) and content checksums with one device
    chain per frame; the host XXH32 is for the header byte only.  lzf_frame_get_stats counts both paths: this test fails
    if a block checksum (or the content checksum of a frame this small) is ever computed on the host."""
    from rust_lz_fear_amd import ffi
    data = [synth.silesia_mix((70 + 3 * k) << 20, ((70 + 3 * k) << 20) + 700_000 + 1111 * k).tobytes() for k in range(5)]
    data.append(vectors.rng_bytes(5, 200_000))                            # stored blocks: their checksums are over the raw bytes
    for kw in (dict(block_size=64 << 10, block_checksums=True), dict(block_size=64 << 10, block_checksums=True, independent_blocks=False),
               dict(block_size=256 << 10, block_checksums=True, dictionary=synth.gen_text_zipf(8, 30000).tobytes())):
        g, okw = settings_pair(**kw)
        nblk = sum((len(d) + kw["block_size"] - 1) // kw["block_size"] for d in data)
        s0 = ffi.frame_stats()
        frames = g.compress_many(data)
        s1 = ffi.frame_stats()
        assert s1["host_block_hashes"] == s0["host_block_hashes"]
        assert s1["device_block_hashes"] - s0["device_block_hashes"] == nblk
        dev_c, host_c = s1["device_content_hashes"] - s0["device_content_hashes"], s1["host_content_hashes"] - s0["host_content_hashes"]
        assert dev_c + host_c == len(data) and (host_c == 0 or "dictionary" in kw)      # dict ++ block layouts hash the (non-contiguous) frame on the workers
        for d, f in zip(data, frames):
            assert f == o.frame_compress(d, o.make_settings(**okw))[1]
        res = framed.decompress_frames(frames, dictionary=kw.get("dictionary", b""))
        s2 = ffi.frame_stats()
        assert [r for r in res] == [(0, d) for d in data]
        assert s2["host_block_hashes"] == s0["host_block_hashes"] and s2["host_content_hashes"] == s1["host_content_hashes"]
        assert s2["device_block_hashes"] - s1["device_block_hashes"] == nblk
        assert s2["device_content_hashes"] - s1["device_content_hashes"] == len(data)
        # a flipped payload byte / checksum byte: BlockChecksumFail at that block, blocks before it delivered, same `consumed`
        f = bytearray(frames[0]); f[len(f) // 2] ^= 0x40
        erc, eout, eused = o.frame_decompress(bytes(f), dictionary=kw.get("dictionary", b""), cap=8 << 20)
        (rc, out, used), = framed.decompress_frames([bytes(f)], dictionary=kw.get("dictionary", b""), caps=[8 << 20], with_consumed=True)
        assert (rc, out, used) == (erc, eout, eused) and rc == 19


def test_consumed_matches_the_reference_reader_on_malformed_frames():
    """*consumed = what the reference's reader has read from the input when it stops (errors included)."""
    rng = np.random.default_rng(4242)
    data = synth.silesia_mix(90 << 20, (90 << 20) + 400_000).tobytes()
    frames = []
    for kw in (dict(block_size=64 << 10), dict(block_size=64 << 10, independent_blocks=False, block_checksums=True), dict(block_size=64 << 10, block_checksums=True)):
        base = o.frame_compress(data, o.make_settings(**kw))[1]
        frames += [base, base + b"trailing bytes"] + [mutate(rng, base) for _ in range(30)]
    exp = [o.frame_decompress(f, cap=8 << 20) for f in frames]
    got = framed.decompress_frames(frames, caps=[8 << 20] * len(frames), with_consumed=True)
    for (erc, eout, eused), (rc, out, used) in zip(exp, got):
        assert (rc, out) == (erc, eout)
        assert used == eused, (rc, used, eused)


def test_staging_moves_whole_pieces_not_blocks():
    """Host staging: one pinned slab, one asynchronous DMA per 4 MiB piece — not one copy per block.  16384 blocks of 4 KiB
    must not take 16384 host-to-device copies."""
    from rust_lz_fear_amd import ffi
    data = synth.silesia_mix(5 << 20, (5 << 20) + (24 << 20)).tobytes()
    n = 6000
    blocks = [data[i * 4096:(i + 1) * 4096] for i in range(n)]
    s0 = ffi.frame_stats()
    comp = ffi.compress_blocks_host([dict(input=b, out_cap=len(b) + 64) for b in blocks])
    s1 = ffi.frame_stats()
    assert all(rc == 0 for rc, _ in comp)
    assert s1["h2d_copies"] - s0["h2d_copies"] <= 16 and s1["d2h_copies"] - s0["d2h_copies"] <= 16
    dec = ffi.decompress_blocks_host([dict(input=c, limit=4096, out_cap=4096 + len(c) + 64) for _, c in comp])
    s2 = ffi.frame_stats()
    assert [d for _, d in dec] == blocks
    assert s2["h2d_copies"] - s1["h2d_copies"] <= 16 and s2["d2h_copies"] - s1["d2h_copies"] <= 16
    assert s2["pinned_bytes"] >= 24 << 20
    # frames: 40 frames of 1.5 MiB at 64 KiB blocks = 960 blocks, a handful of DMAs each way
    frames_in = [data[k * 1_500_000:(k + 1) * 1_500_000] for k in range(12)]
    g, okw = settings_pair(block_size=64 << 10)
    fr = g.compress_many(frames_in)
    s3 = ffi.frame_stats()
    assert s3["h2d_copies"] - s2["h2d_copies"] <= 12 and s3["d2h_copies"] - s2["d2h_copies"] <= 12
    assert [r for r in framed.decompress_frames(fr)] == [(0, d) for d in frames_in]
    ffi.lib().lzf_frame_release_scratch()
    assert ffi.frame_stats()["pinned_bytes"] == 0
    assert framed.decompress_frames(fr[:2]) == [(0, d) for d in frames_in[:2]]         # scratch comes back on demand


def test_decompress_many_memory_budget_slices_and_refuses():
    """a frame's device footprint is bounded by what its blocks can expand to (255 x compressed size), not by
    block_maxsize per block; lzf_frame_decompress_many works through the frames in as many passes as its memory budget
    needs, and a frame that does not fit alone reports LZF_E_NO_MEMORY without taking the others down."""
    from rust_lz_fear_amd import ffi
    data = [synth.silesia_mix((100 + k) << 20, ((100 + k) << 20) + 900_000).tobytes() for k in range(9)]
    frames = [o.frame_compress(d, o.make_settings(block_size=64 << 10, independent_blocks=bool(k % 2)))[1] for k, d in enumerate(data)]
    big = o.frame_compress(synth.silesia_mix(0, 40 << 20).tobytes(), o.make_settings(block_size=1 << 20))[1]
    # 20000 five-byte blocks that claim a 4 MiB block size: 80 GB under the old accounting, a few MB now
    tiny_blk = bytes([0x10, 0x61, 0x01, 0x00]) + b""                      # 1 literal 'a', match len 4 offset 1 -> "aaaaa"
    many = bytes.fromhex("04224d186070") + bytes([0x73]) + b"".join(len(tiny_blk).to_bytes(4, "little") + tiny_blk for _ in range(20000)) + bytes(4) + (0).to_bytes(4, "little")
    erc, eout, _ = o.frame_decompress(many, cap=1 << 20)
    try:
        ffi.lib().lzf_frame_set_memory_budget(64 << 20)
        res = framed.decompress_frames(frames + [big] + frames[:2] + [many], caps=[2 << 20] * 9 + [48 << 20] + [2 << 20] * 2 + [1 << 20])
    finally:
        ffi.lib().lzf_frame_set_memory_budget(0)
    assert res[:9] == [(0, d) for d in data] and res[10:12] == [(0, d) for d in data[:2]]
    assert res[9][0] == ffi.E_NO_MEMORY and res[9][1] == b""
    assert res[12] == (erc, eout)
    assert framed.decompress_frames([big], caps=[48 << 20])[0][0] == 0                  # fits under the default budget


def test_many_frames_through_a_pinned_ring_smaller_than_the_call():
    """host_staging.cpp: a pass that moves more than the pinned slab may hold goes through it as a ring of 4 MiB slots (a slot is
    reused when the DMA that last read it has finished).  With the slab limited to 12 MiB — three slots — 24 frames (72 MiB in,
    every payload and every decoded byte out) make the ring go round many times in both directions (and once more with a 16 MiB limit, where
    the slab's growth step of 64 MiB used to carry it over the limit): frames == the oracle's,
    round trip exact, linked frames and dictionary frames (their own layouts of the slab) included, small frames (the inline
    path borrows a slot too) in between, and the pinned footprint stays at the limit."""
    from rust_lz_fear_amd import ffi
    datas = [synth.silesia_mix((7 * k) << 20, ((7 * k) << 20) + (3 << 20) + 4099 * k).tobytes() for k in range(20)] + [b"", b"abc" * 50, synth.silesia_mix(0, 70_000).tobytes(), b"x" * 300_000]
    dic = synth.silesia_mix(1 << 20, (1 << 20) + 20_000).tobytes()
    ffi.lib().lzf_frame_release_scratch()
    ffi.lib().lzf_frame_set_pinned_limit(12 << 20)
    try:
        for kw in (dict(block_size=256 << 10), dict(block_size=64 << 10, independent_blocks=False), dict(block_size=1 << 20, block_checksums=True, dictionary=dic)):
            so = o.make_settings(**{k: v for k, v in kw.items() if k != "dictionary"}, **({"dictionary": dic, "dictionary_id": 0} if "dictionary" in kw else {}))
            want = [o.frame_compress(d, so)[1] for d in datas]
            cs = framed.CompressionSettings().block_size(kw["block_size"]).independent_blocks(kw.get("independent_blocks", True)).block_checksums(kw.get("block_checksums", False))
            if "dictionary" in kw:
                cs = cs.dictionary(0, dic)
            got = cs.compress_many(datas)
            assert got == want, kw
            back = framed.decompress_frames(got, dictionary=kw.get("dictionary", b""), caps=[len(d) + 64 for d in datas])
            assert back == [(0, d) for d in datas], kw
            assert 0 < ffi.frame_stats()["pinned_bytes"] <= 12 << 20
        ffi.lib().lzf_frame_set_pinned_limit(16 << 20)
        cs = framed.CompressionSettings().block_size(256 << 10)
        assert cs.compress_many(datas) == [o.frame_compress(d, o.make_settings(block_size=256 << 10))[1] for d in datas]
        assert 0 < ffi.frame_stats()["pinned_bytes"] <= 16 << 20
    finally:
        ffi.lib().lzf_frame_set_pinned_limit(0)
        ffi.lib().lzf_frame_release_scratch()


def test_c_abi_block_reader_is_decode_block():
    """lzf_frame_reader_* == LZ4FrameReader::new + decode_block, block by block: same blocks, same error kind at the same
    block, same bytes consumed, dictionary and carried window included."""
    data = synth.silesia_mix(33 << 20, (33 << 20) + 700_000).tobytes()
    dic = synth.gen_text_zipf(12, 40000).tobytes()
    rng = np.random.default_rng(99)
    cases = []
    for kw in (dict(block_size=64 << 10), dict(block_size=64 << 10, independent_blocks=False), dict(block_size=256 << 10, block_checksums=True),
               dict(block_size=64 << 10, independent_blocks=False, dictionary=dic, dictionary_id=3), dict(block_size=64 << 10, dictionary=dic, dictionary_id=3)):
        f = o.frame_compress(data, o.make_settings(**kw))[1]
        cases.append((f, kw.get("dictionary", b"")))
        cases += [(mutate(rng, f), kw.get("dictionary", b"")) for _ in range(6)]
    cases.append((o.frame_compress(vectors.rng_bytes(3, 150_000), o.make_settings(block_size=64 << 10))[1], b""))       # stored blocks
    cases.append((o.frame_compress(b"", o.make_settings())[1], b""))
    for f, d in cases:
        erc, eout, eused = o.frame_decompress(f, dictionary=d, cap=8 << 20)
        got, rc, r = b"", 0, None
        try:
            r = framed.FrameBlockReader(f)
            while not r.finished():
                b = r.decode_block(d)
                got += b
                if not b and not r.finished():
                    break                                     # a block of zero bytes: read_to_end stops here (:52-71,:286)
                assert len(b) <= r.info.block_maxsize
        except framed.FrameError as e:
            rc = e.code
        assert rc == erc, (rc, erc)
        assert got == eout
        if r is not None:
            assert r.consumed() == eused, (rc, r.consumed(), eused)
    hdr_err = bytearray(cases[0][0]); hdr_err[0] ^= 1
    with pytest.raises(framed.FrameError) as e:
        framed.FrameBlockReader(bytes(hdr_err))
    assert e.value.code == 17


def test_examples_dolz4_delz4_round_trip(tmp_path):
    """tools/dolz4.py / tools/delz4.py (the reference's examples/dolz4.rs, delz4.rs): file -> .lz4 -> file, and the .lz4 is
    the frame the reference writes for the same settings."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = synth.silesia_mix(20 << 20, (20 << 20) + 9_000_000).tobytes()
    src = tmp_path / "input.bin"; src.write_bytes(data)
    lz = tmp_path / "input.bin.lz4"; back = tmp_path / "back.bin"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dolz4.py"), str(src), str(lz)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    frame = lz.read_bytes()
    assert frame == o.frame_compress(data, o.make_settings(content_size=len(data)))[1]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "delz4.py"), str(lz), str(back)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert back.read_bytes() == data


def test_examples_stream_a_file_larger_than_the_memory_budget(tmp_path):
    """The drivers stream like the originals (examples/dolz4.rs:10-17: File -> File through compress_with_size; examples/delz4.rs:31-38:
    fill_buf / consume): a 128 MiB file with the device-memory budget of the frame layer set to 16 MiB — eight times smaller — goes
    through dolz4 (4 MiB pieces into the frame writer) and delz4 (four blocks read ahead) with the Python side never holding more than
    a fraction of the file (tracemalloc peak), the .lz4 is the oracle's frame and the round trip is exact."""
    import importlib.util
    import tracemalloc
    from rust_lz_fear_amd import ffi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "tools", name + ".py"))
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
        return m
    dolz4, delz4 = load("dolz4"), load("delz4")
    total = 128 << 20
    data = synth.silesia_mix(0, total).tobytes()
    src = tmp_path / "big.bin"; src.write_bytes(data)
    lz = tmp_path / "big.bin.lz4"; back = tmp_path / "big.back"
    want = o.frame_compress(data, o.make_settings(content_size=len(data)))[1]
    del data
    ffi.lib().lzf_frame_set_memory_budget(16 << 20)
    try:
        tracemalloc.start()
        n_in, n_out = dolz4.compress_file(framed.CompressionSettings(), str(src), str(lz), with_size=True, piece=4 << 20)
        peak_c = tracemalloc.get_traced_memory()[1]; tracemalloc.reset_peak()
        m_in, m_out = delz4.decompress_file(str(lz), str(back), readahead=4)
        peak_d = tracemalloc.get_traced_memory()[1]
        tracemalloc.stop()
    finally:
        ffi.lib().lzf_frame_set_memory_budget(0)
    assert (n_in, m_out) == (total, total) and n_out == m_in == len(want)
    assert lz.read_bytes() == want
    assert back.read_bytes() == src.read_bytes()
    # bounded by the pieces / blocks in flight — per block read ahead: its compressed bytes, its output slot (block_maxsize + compressed
    # size, the out_cap of exact malformed-input parity) and the decoded block — not by the file
    assert peak_c < 16 << 20 and peak_d < 12 * (4 << 20), (peak_c, peak_d)


@pytest.mark.parametrize("kw", [dict(block_size=64 << 10, block_checksums=True), dict(block_size=64 << 10, dictionary=synth.gen_text_zipf(21, 5000).tobytes(), dictionary_id=5)])
def test_compress_many_in_pipelined_groups_is_still_the_reference_frame(kw):
    """More than 9216 independent blocks in one lzf_frame_compress_many call travel in several groups (own device buffers,
    upload / compress / download overlapping): every frame must still be the oracle's frame, checksums included, with an empty
    input, a stored-block input and ragged sizes among them; and the frames decode back in one call."""
    from rust_lz_fear_amd import ffi
    mix = synth.silesia_mix(0, 200 << 20)
    sizes = [27 << 20] * 10 + [0, (27 << 20) + 12345, 5, (26 << 20) + 65536 * 3 + 1] + [28 << 20] * 9
    datas, pos = [], 0
    for i, n in enumerate(sizes):
        datas.append(mix[pos % (mix.size - n): pos % (mix.size - n) + n].tobytes() if n else b""); pos += n + 777_777
    datas.append(vectors.rng_bytes(17, 3 << 20))
    assert sum((len(d) + 65535) // 65536 for d in datas) > 9216
    g, okw = settings_pair(**kw)
    s0 = ffi.frame_stats()
    frames = g.compress_many(datas)
    s1 = ffi.frame_stats()
    es = o.make_settings(**okw)
    for d, f in zip(datas, frames):
        assert f == o.frame_compress(d, es)[1], len(d)
    if kw.get("block_checksums"):
        assert s1["host_block_hashes"] == s0["host_block_hashes"]
        assert s1["device_block_hashes"] - s0["device_block_hashes"] == sum((len(d) + 65535) // 65536 for d in datas)
    back = framed.decompress_frames(frames, dictionary=kw.get("dictionary", b""), caps=[len(d) + 64 for d in datas])
    assert back == [(0, d) for d in datas]


def test_frames_written_by_liblz4_decode_on_the_gpu_and_ours_in_liblz4():
    """Frame-level interop with the C implementation (fuzz/fuzz_targets/interop_decode.rs:6-31): the committed LZ4F_compressFrame
    frames (tests/golden/lz4f_frames.*: HC levels, linked blocks, block checksums, content size, stored blocks) decode on the GPU —
    whole-frame driver, block-by-block C reader and the streaming Python reader — and, when liblz4 is on the box, LZ4F_decompress
    reads the GPU's frames.  An independent pin of the frame layer: liblz4 shares no code with this repository."""
    import io
    import liblz4_ffi as c
    J = json.load(open(os.path.join(GOLD, "lz4f_frames.json")))
    blob = open(os.path.join(GOLD, "lz4f_frames.bin"), "rb").read()
    frames, datas = [], []
    for fr in J["frames"]:
        frames.append(blob[fr["offset"]: fr["offset"] + fr["length"]])
        datas.append(eval(fr["input"], {"synth": synth}).tobytes())
        assert [len(datas[-1]), "%08x" % o.xxh32(datas[-1])] == fr["in"]
    got = framed.decompress_frames(frames, caps=[len(d) + 64 for d in datas], with_consumed=True)
    assert got == [(0, d, len(f)) for d, f in zip(datas, frames)]
    for f, d in zip(frames, datas):
        r = framed.FrameBlockReader(f)
        out = b""
        while not r.finished():
            out += r.decode_block()
        assert out == d and r.consumed() == len(f)
        assert framed.LZ4FrameReader(io.BytesIO(f), readahead=3).read() == d
    if c.available():
        data = synth.silesia_mix(10 << 20, (10 << 20) + 400_000).tobytes()
        for kw in (dict(), dict(block_size=64 << 10), dict(block_size=64 << 10, independent_blocks=False), dict(block_size=256 << 10, block_checksums=True)):
            g, _ = settings_pair(**kw)
            assert c.lz4f_decompress(g.compress(data), len(data) + 64) == (True, data), kw
