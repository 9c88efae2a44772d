"""CPU tests: pin the oracle (oracle/lzf_oracle.c) against everything the reference's own
tests hold for the hot path.  No GPU needed."""
import json
import os

import pytest
import xxhash

import oracle_ffi as o
import liblz4_ffi as c
import vectors
from rust_lz_fear_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def fp(b):
    return [len(b), "%08x" % xxhash.xxh32(b).intdigest()]


def test_xxh32_matches_python_xxhash():
    for n in [0, 1, 3, 4, 15, 16, 17, 31, 32, 33, 100, 1000, 65536]:
        d = vectors.rng_bytes(n, n)
        assert o.xxh32(d) == xxhash.xxh32(d).intdigest()
        assert o.xxh32(d, 7) == xxhash.xxh32(d, seed=7).intdigest()


# ---- src/raw/decompress.rs:153-175
@pytest.mark.parametrize("data,status,expect", vectors.DECODE_KATS)
def test_reference_decode_kats(data, status, expect):
    rc, out = o.decompress_raw(data)
    assert rc == status
    if expect is not None:
        assert out == expect


# ---- src/lib.rs:43-95 (round trips; U16Table iff len <= 0xFFFF, :26-30)
@pytest.mark.parametrize("s", vectors.LIB_RS_STRINGS)
def test_reference_roundtrip_strings(s):
    kind = o.TABLE_U16 if len(s) <= 0xFFFF else o.TABLE_U32
    rc, comp = o.compress2(s, kind=kind)
    assert rc == 0
    rc, dec = o.decompress_raw(comp)
    assert rc == 0 and dec == s


def test_reference_compression_works():
    s = vectors.LIB_RS_STRINGS[-1]
    rc, comp = o.compress2(s, kind=o.TABLE_U16)
    assert rc == 0 and len(comp) < len(s)          # src/lib.rs:89-95


def test_reference_big_compression_shape():
    s = vectors.big_compression_bytes(8_000_000)    # src/lib.rs:98-106 at 1/10 scale
    rc, comp = o.compress2(s)
    assert rc == 0
    rc, dec = o.decompress_raw(comp, cap=len(s) + 64)
    assert rc == 0 and dec == s


def test_empty_input_emits_nothing():               # quirk B4, mod.rs:171
    assert o.compress2(b"") == (0, b"")
    assert o.decompress_raw(b"") == (0, b"")


# ---- SURVEY.md Appendix B/C fingerprints
def test_survey_fingerprints():
    S = json.load(open(os.path.join(GOLD, "survey_fingerprints.json")))
    e = bytearray(synth.lcg_bytes(1, 156)); e[143:149] = e[11:17]
    assert fp(o.compress2(bytes(e))[1]) == S["KAT-A"]["u32_raw"]
    if c.available():
        assert fp(c.compress_fresh_stream(bytes(e))) == S["KAT-A"]["liblz4_fresh_stream"]   # B2 divergence
    g1 = vectors.big_compression_bytes(10 ** 6)
    assert fp(g1) == S["G1"]["in"] and fp(o.compress2(g1)[1]) == S["G1"]["u32_raw"]
    g2 = bytes(65536)
    assert fp(g2) == S["G2"]["in"] and fp(o.compress2(g2)[1]) == S["G2"]["u32_raw"]
    g3 = synth.lcg_bytes(5, 262144, 3)
    assert fp(g3) == S["G3"]["in"] and fp(o.compress2(g3)[1]) == S["G3"]["u32_raw"]
    assert fp(o.frame_compress(g3, o.make_settings(block_size=65536))[1]) == S["G3"]["frame_64k_independent"]
    rc, f = o.frame_compress(g3, o.make_settings(block_size=65536, independent_blocks=False,
                                                 block_checksums=True, content_size=len(g3)))
    assert fp(f) == S["G3"]["frame_64k_linked_blocksum_csize"]
    assert o.frame_decompress(f)[:2] == (0, g3)
    g4 = synth.lcg_bytes(7, 65535, 1)
    assert fp(g4) == S["G4"]["in"]
    assert fp(o.compress2(g4, kind=o.TABLE_U16)[1]) == S["G4"]["u16_raw"]
    assert fp(o.compress2(g4, kind=o.TABLE_U32)[1]) == S["G4"]["u32_raw"]
    kb = synth.lcg_bytes(3, 69632, 3)
    rc, f = o.frame_compress(kb, o.make_settings(block_size=65536, independent_blocks=False))
    assert fp(f) == S["KAT-B"]["frame_64k_linked"]                  # quirks B1/B3
    assert o.frame_decompress(f)[:2] == (0, kb)


def test_survey_fingerprint_g5_4mib():
    S = json.load(open(os.path.join(GOLD, "survey_fingerprints.json")))
    g5 = synth.lcg_bytes(9, 4 << 20, 7)
    assert fp(g5) == S["G5"]["in"]
    assert fp(o.compress2(g5)[1]) == S["G5"]["u32_raw"]
    rc, f = o.frame_compress(g5)
    assert fp(f) == S["G5"]["frame_default"]
    assert o.frame_decompress(f)[:2] == (0, g5)


# ---- tests/issue-15.rs: linked 64 KiB blocks round trip
def test_issue15_regression():
    data = open(os.path.join(GOLD, "issue15_input.bin"), "rb").read()
    assert len(data) == 81248
    rc, f = o.frame_compress(data, o.make_settings(independent_blocks=False, block_size=64 * 1024))
    assert rc == 0 and len(f) == 81160
    rc, dec, used = o.frame_decompress(f)
    assert rc == 0 and dec == data and used == len(f)


# ---- fuzz corpus frames (self-checking content checksums)
def test_corpus_frame_uncomp_data():
    data = open(os.path.join(GOLD, "uncomp.data.lz4"), "rb").read()
    J = json.load(open(os.path.join(GOLD, "corpus_frames.json")))["valid_frames"]["uncomp.data.lz4"]
    rc, dec, used = o.frame_decompress(data)
    assert rc == 0 and used == len(data)
    assert fp(dec) == [J["out_len"], J["out_xxh32"]]
    # corrupting one payload byte must trip the content checksum
    bad = bytearray(data); bad[20] ^= 1
    assert o.frame_decompress(bytes(bad))[0] == o.F_FRAME_CHECKSUM_FAIL


@pytest.mark.skipif(not os.path.isdir("/root/reference/fuzz/corpus/decode"), reason="reference tree not mounted")
def test_decode_corpus_census_matches_fixture():
    """The 1022-file malformed-input corpus: error class per file is pinned by the fixture, the
    class totals relate to SURVEY.md §4 (see tests/golden/README.md for the 10 reclassified files)."""
    import collections, glob
    J = json.load(open(os.path.join(GOLD, "corpus_frames.json")))
    census = collections.Counter()
    for f in sorted(glob.glob("/root/reference/fuzz/corpus/decode/*")):
        data = open(f, "rb").read()
        rc, dec, _ = o.frame_decompress(data)
        assert o.STATUS_NAMES[rc] == J["per_file"][os.path.basename(f)]
        census[o.STATUS_NAMES[rc]] += 1
        if os.path.basename(f) in J["valid_frames"]:
            v = J["valid_frames"][os.path.basename(f)]
            assert rc == 0 and fp(dec) == [v["out_len"], v["out_xxh32"]]
    assert dict(census) == J["census"]
    assert sum(census.values()) == 1022 and census["WrongMagic"] == 661
    # SURVEY.md §4 census, with "a block that decodes to 0 bytes ends read_to_end" applied:
    survey = {"Ok": 4, "WrongMagic": 661, "InputError": 99, "BlockSizeOverflow": 92,
              "InvalidDeduplicationOffset": 85, "UnexpectedEnd": 33, "ZeroDeduplicationOffset": 18,
              "BlockChecksumFail": 12, "FrameChecksumFail": 7, "ReservedFlagBitsSet": 3,
              "HeaderChecksumFail": 3, "UnimplementedBlocksize": 2, "UnsupportedVersion": 1,
              "MemoryLimitExceeded": 1, "ReservedBdBitsSet": 1}
    diff = {k: census.get(k, 0) - survey[k] for k in survey if census.get(k, 0) != survey[k]}
    assert diff == {"Ok": 10, "InputError": -8, "BlockSizeOverflow": -2}


@pytest.mark.skipif(not os.path.isdir("/root/reference/fuzz/corpus"), reason="reference tree not mounted")
def test_fuzz_plaintext_corpora_roundtrip_and_match_c():
    """interop_decode + roundtrip_fuzz corpora are 501 plaintext inputs (fuzz_targets/*.rs)."""
    import glob
    files = sorted(glob.glob("/root/reference/fuzz/corpus/interop_decode/*") +
                   glob.glob("/root/reference/fuzz/corpus/roundtrip_fuzz/*"))
    assert len(files) == 501
    for f in files:
        data = open(f, "rb").read()
        rc, frame = o.frame_compress(data)          # roundtrip_fuzz.rs:10-14
        assert rc == 0
        assert o.frame_decompress(frame)[:2] == (0, data)
        if c.available() and 0 < len(data) <= 0xFFFF:
            assert o.compress2(data, kind=o.TABLE_U16)[1] == c.compress_default(data)


# ---- liblz4 equality regimes (committed vectors; live cross-check when the library exists)
def test_liblz4_golden_vectors():
    J = json.load(open(os.path.join(GOLD, "liblz4_vectors.json")))
    cases = dict(vectors.small_cases() + vectors.medium_cases() +
                 [(f"librs{i}", s) for i, s in enumerate(vectors.LIB_RS_STRINGS)])
    assert len(J["u32"]) >= 200 and len(J["u16"]) >= 200
    for kind, key in ((o.TABLE_U32, "u32"), (o.TABLE_U16, "u16")):
        for name, v in J[key].items():
            data = cases[name]
            assert fp(data) == v["in"], name
            rc, comp = o.compress2(data, kind=kind)
            assert rc == 0
            assert (fp(comp) == v["c"]) == v["equal_expected"], (key, name)


@pytest.mark.skipif(not c.available(), reason="liblz4 not installed")
def test_liblz4_decodes_oracle_output_and_vice_versa():
    for name, data in vectors.medium_cases():
        rc, comp = o.compress2(data)
        n, dec = c.decompress_safe(comp, len(data))
        assert n == len(data) and dec == data, name
        assert o.decompress_raw(c.compress_default(data), cap=len(data) + 64) == (0, data)


# ---- frame layer: flag matrix of tests/output_equivalence.rs (round trip; exact bytes vs C are
#      pinned through the raw-block vectors above)
@pytest.mark.parametrize("bits", range(32))
def test_frame_flag_matrix_roundtrip(bits):
    data = synth.silesia_mix(0, 300_000).tobytes()
    dict_data = bytes([1, 3, 3, 7])
    kw = dict(content_checksum=not (bits & 1), independent_blocks=not (bits & 2),
              block_size=(256 << 10) if bits & 4 else (64 << 10))
    if bits & 8:
        kw["dictionary"] = dict_data            # dictionary(0, d).dictionary_id_nonsense_override(None)
    if bits & 16:
        kw["content_size"] = len(data)
    rc, f = o.frame_compress(data, o.make_settings(**kw))
    assert rc == 0
    rc, dec, used = o.frame_decompress(f, dictionary=dict_data if bits & 8 else b"")
    assert rc == 0 and dec == data and used == len(f)


def test_frame_dictionary_linked_and_independent():
    d = synth.gen_text_zipf(3, 70000).tobytes()
    data = synth.gen_text_zipf(4, 200000).tobytes()
    for indep in (True, False):
        s = o.make_settings(independent_blocks=indep, block_size=64 << 10, dictionary=d, dictionary_id=42)
        rc, f = o.frame_compress(data, s)
        assert rc == 0
        assert o.frame_decompress(f, dictionary=d)[:2] == (0, data)
        rc2 = o.frame_decompress(f)[0]
        assert rc2 != 0      # without the dictionary the frame must not decode cleanly


def test_frame_block_size_validation():
    # header.rs:53-62 via framed/compress.rs:183
    for bs in (64 << 10, 256 << 10, 1 << 20, 4 << 20):
        assert o.frame_compress(b"x" * 100, o.make_settings(block_size=bs))[0] == 0
    for bs in (1, 3, 1000, 32 << 10, 128 << 10, 2 << 20, 8 << 20, (64 << 10) + 1):
        assert o.frame_compress(b"x", o.make_settings(block_size=bs))[0] == o.F_INVALID_BLOCK_SIZE
    for bs in (0, 16 << 20, 32 << 20):   # BlockDescriptor::new unwrap() panics (header.rs:55)
        assert o.frame_compress(b"x", o.make_settings(block_size=bs))[0] == o.F_PANIC


def test_frame_incompressible_blocks_are_stored():
    data = vectors.rng_bytes(5, 200000)
    rc, f = o.frame_compress(data, o.make_settings(block_size=64 << 10))
    assert rc == 0 and len(f) == 7 + 4 * 4 + len(data) + 4 + 4      # header, 4 block words, EndMark, checksum
    assert o.frame_decompress(f)[:2] == (0, data)


def test_output_full_and_cap_n_quirk():
    # NoPartialWrites cap (framed/compress.rs:242): C <= N passes, else OutputFull
    data = vectors.rng_bytes(11, 5000)
    assert o.compress2(data, cap=len(data))[0] == o.OUTPUT_FULL
    rc, comp = o.compress2(data)
    assert rc == 0 and len(comp) > len(data)
    assert o.compress2(data, cap=len(comp))[0] == 0
    assert o.compress2(data, cap=len(comp) - 1)[0] == o.OUTPUT_FULL


def test_u16_table_contract():
    assert o.compress2(bytes(65536), kind=o.TABLE_U16)[0] == o.CONTRACT      # mod.rs:167 assert
    assert o.compress2(bytes(65535), kind=o.TABLE_U16)[0] == 0


def test_prefix_and_table_carry_equals_linked_frame_blocks():
    """compress2 with cursor > 0 and a carried table == what the frame layer does (A.2)."""
    data = synth.silesia_mix(20 << 20, (20 << 20) + 200000).tobytes()
    bs = 65536
    t = o.new_table()
    buf = b""
    pieces = []
    for off in range(0, len(data), bs):
        blk = data[off:off + bs]
        inp = buf + blk
        rc, comp = o.compress2(inp, cursor=len(buf), table=t, cap=len(blk))
        pieces.append((rc, comp, blk))
        buf = inp
        if len(buf) > 65536:
            forget = len(buf) - 65536
            t.offset += forget
            buf = buf[forget:]
    rc, f = o.frame_compress(data, o.make_settings(independent_blocks=False, block_size=bs, content_checksum=False))
    assert rc == 0
    body = b""
    for rc, comp, blk in pieces:
        if rc == 0:
            body += len(comp).to_bytes(4, "little") + comp
        else:
            body += (len(blk) | 0x80000000).to_bytes(4, "little") + blk
    assert f[7:-4] == body


def test_hc_fixtures_decode_with_the_oracle():
    """tests/golden/hc_blocks.*: blocks from liblz4's HC / fast encoders decode to the regenerated inputs."""
    from rust_lz_fear_amd import synth
    J = json.load(open(os.path.join(GOLD, "hc_blocks.json")))
    blob = open(os.path.join(GOLD, "hc_blocks.bin"), "rb").read()
    assert len(J["blocks"]) == 18
    for b in J["blocks"]:
        comp = blob[b["offset"]: b["offset"] + b["length"]]
        assert [len(comp), "%08x" % o.xxh32(comp)] == b["comp"]
        data = eval(b["input"], {"synth": synth}).tobytes()
        assert [len(data), "%08x" % o.xxh32(data)] == b["in"]
        assert o.decompress_raw(comp, limit=len(data)) == (0, data)


def test_lz4f_fixture_frames_decode_with_the_oracle():
    """tests/golden/lz4f_frames.*: frames written by liblz4's own frame layer (LZ4F_compressFrame, levels 0 / 4 / 9: the reference's
    interop_decode fuzz target) decode to the regenerated inputs, every byte of the frame consumed."""
    J = json.load(open(os.path.join(GOLD, "lz4f_frames.json")))
    blob = open(os.path.join(GOLD, "lz4f_frames.bin"), "rb").read()
    assert len(J["frames"]) == 8
    for fr in J["frames"]:
        frame = blob[fr["offset"]: fr["offset"] + fr["length"]]
        assert [len(frame), "%08x" % o.xxh32(frame)] == fr["frame"]
        data = eval(fr["input"], {"synth": synth}).tobytes()
        assert [len(data), "%08x" % o.xxh32(data)] == fr["in"]
        assert o.frame_decompress(frame, cap=len(data) + 64) == (0, data, len(frame))


@pytest.mark.skipif(not c.available(), reason="liblz4 not installed")
def test_liblz4_frame_layer_reads_oracle_frames():
    """The other direction, live: LZ4F_decompress accepts the oracle's frames over the flag matrix (the frame FORMAT is pinned by an
    implementation that shares no code with this repository)."""
    data = synth.silesia_mix(10 << 20, (10 << 20) + 400_000).tobytes()
    for kw in (dict(), dict(block_size=64 << 10), dict(block_size=64 << 10, independent_blocks=False), dict(block_size=256 << 10, block_checksums=True),
               dict(block_size=64 << 10, content_checksum=False, block_checksums=True, independent_blocks=False), dict(content_size=len(data))):
        f = o.frame_compress(data, o.make_settings(**kw))[1]
        assert c.lz4f_decompress(f, len(data) + 64) == (True, data), kw
