"""The team compress kernel's SOURCE (rust-lz-fear_amd/csrc/lz4_compress_team.inc — the latency class of lzf_compress_batch: a searcher,
an emitter and a feeder wavefront per block, input ring and position table in LDS), compiled for the CPU against the lock-step
wavefront emulator of lzf_simt.h, must produce the oracle's bytes — raw::compress2 (src/raw/compress/mod.rs:165-238) with a fresh
U32Table or a read-only template.  No GPU: this is the kernel's parse / commit logic (fast batches on the LDS ring, the wave-wide
match measurement, general batches, the skip schedule, extension, backtrack), its descriptor hand-over, the emitter's write_group /
cap = N refusals and the feeder's ring protocol (claims, skips over long matches, the 128 KiB wrap) checked on the build host under
one fixed interleaving of the three waves; the -m gpu tests check the same source as compiled by hipcc, on the real scheduler.
"""
import ctypes as C
import random

import numpy as np
import pytest

import emu_ffi
import oracle_ffi as o
import vectors
from rust_lz_fear_amd import synth


def run(inputs, cursors=None, caps=None, tables=None, **kw):
    res, _ = emu_ffi.compress_batch(inputs, cursors=cursors, caps=caps, tables=tables, kernel="team", **kw)
    return res


def expect(inputs, res, cursors=None, caps=None, otables=None, names=None):
    for i, d in enumerate(inputs):
        t = None
        if otables is not None and otables[i] is not None:
            t = o.new_table(); C.memmove(C.addressof(t), C.addressof(otables[i]), C.sizeof(t))
        erc, eout = o.compress2(d, cursor=cursors[i] if cursors else 0, cap=caps[i] if caps else None, table=t)
        rc, out = res[i]
        name = names[i] if names else i
        assert rc == erc, (name, rc, erc)
        if erc == 0:
            assert out == eout, (name, len(out), len(eout))
        # (a refused job's out_len is unspecified by the ABI: the sink is dropped, framed/compress.rs:250-255)


def test_small_and_boundary_inputs():
    cases = vectors.small_cases() + [(f"lib{i}", s) for i, s in enumerate(vectors.LIB_RS_STRINGS)]
    inputs = [d for _, d in cases]
    expect(inputs, run(inputs), names=[n for n, _ in cases])


def test_survey_fingerprints_and_quirks():
    e = bytearray(synth.lcg_bytes(1, 156)); e[143:149] = e[11:17]                    # KAT-A (B2)
    inputs = [bytes(e), bytes(65536), synth.lcg_bytes(5, 262144, 3), synth.lcg_bytes(7, 65535, 1), vectors.big_compression_bytes(300000)]
    res = run(inputs)
    expect(inputs, res)
    fp = lambda b: (len(b), "%08x" % o.xxh32(b))
    assert fp(res[0][1]) == (155, "ea8b9d33")
    assert fp(res[1][1]) == (267, "277289eb")
    assert fp(res[2][1]) == (160116, "e26250f2")
    assert fp(res[3][1]) == (60842, "6df079bb")


@pytest.mark.parametrize("part", range(3))
def test_medium_corpus_pieces(part):
    cases = vectors.medium_cases()
    cases = [(n, d[: 320 << 10]) for n, d in cases][part::3]      # the 128 KiB ring wraps twice; candidates up to 64 KiB back
    inputs = [d for _, d in cases]
    expect(inputs, run(inputs), names=[n for n, _ in cases])


def test_incompressible_and_skip_schedule():
    # random bytes: the stride of the skip schedule grows (mod.rs:225-231), every batch is 16 schedule positions; the last
    # literals path (:178-190) carries the whole block; with a repeated piece far into the run B2's probes can still hit
    r = vectors.rng_bytes(5, 200000)
    inputs = [r, r[:70000] + r[1000:1400] + r[70000:90000], r[:30000] + bytes(5000) + r[30000:60000] + r[29000:31000]]
    expect(inputs, run(inputs))


def test_long_matches_and_literal_runs():
    t = synth.gen_text_zipf(9, 5000).tobytes()
    inputs = [bytes(300000),                                             # one match of ~300 000 (length tail of > 1 000 bytes, epochs skipped)
              t + t * 30,                                                # long matches at distance 5 000
              vectors.rng_bytes(3, 3000) + t[:100] + vectors.rng_bytes(4, 70000) + t[:100] * 3,      # literal runs of thousands (COPY state)
              b"ab" * 40000, b"abc" * 30000, b"a" * 17 + b"b" * 100000,
              bytes(range(256)) * 700]
    expect(inputs, run(inputs))


def test_cap_refusals_no_partial_writes():
    d = vectors.rng_bytes(11, 5000)
    t = synth.gen_text_zipf(5, 6000).tobytes()
    _, full_d = o.compress2(d)
    _, full_t = o.compress2(t)
    inputs, caps = [], []
    for data, full in ((d, full_d), (t, full_t)):
        for cap in (len(data), len(full), len(full) - 1, 0, 1, 17, len(full) // 2, len(full) // 2 + 1):
            inputs.append(data); caps.append(cap)
    inputs.append(b""); caps.append(0)
    res = run(inputs, caps=caps)
    expect(inputs, res, caps=caps)
    rng = random.Random(7)
    caps2 = [rng.randrange(0, len(full_t) + 3) for _ in range(40)]
    expect([t] * 40, run([t] * 40, caps=caps2), caps=caps2)


def test_cursor_and_prefix():
    data = synth.silesia_mix(20 << 20, (20 << 20) + 200000).tobytes()
    inputs = [data[:150000], data[:150000], data[:150000], data[:70000], data[:5000]]
    cursors = [65536, 65537, 100, 69990, 5000]       # (cursor > len is the general kernel's job: not "compact")
    expect(inputs, run(inputs, cursors=cursors), cursors=cursors)


def test_dictionary_template_table():
    dic = synth.gen_text_zipf(3, 70000).tobytes()
    tmpl = o.new_table()
    contract = C.c_int(0)
    rep = o.lib().lzfo_u32_replace
    rep.restype = C.c_size_t
    rep.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
    for off in range(0, len(dic) - 7, 3):
        rep(C.addressof(tmpl), dic, len(dic), off, C.byref(contract))
    blocks = [synth.gen_text_zipf(40 + i, 65536).tobytes() for i in range(3)]
    inputs = [dic + b for b in blocks]
    cursors = [len(dic)] * 3
    tabs = [tmpl] * 3
    res = run(inputs, cursors=cursors, tables=tabs, caps=[len(b) for b in blocks])
    expect(inputs, res, cursors=cursors, caps=[len(b) for b in blocks], otables=tabs)


def test_launch_order():
    rng = random.Random(3)
    inputs = [synth.gen_text_zipf(100 + i, rng.randrange(1, 9000)).tobytes() for i in range(23)]
    perm = list(range(23)); rng.shuffle(perm)
    expect(inputs, run(inputs, perm=perm))


def test_random_structures():
    rng = random.Random(11)
    inputs = []
    for k in range(24):
        n = rng.choice([100, 1000, 5000, 20000, 70000, 140000])
        buf = bytearray()
        while len(buf) < n:
            c = rng.random()
            if c < 0.3: buf += vectors.rng_bytes(rng.randrange(1 << 20), rng.randrange(1, 300))
            elif c < 0.5: buf += bytes([rng.randrange(256)]) * rng.randrange(1, 400)
            elif c < 0.8 and len(buf) > 8:
                dist = rng.randrange(1, min(len(buf), 70000) + 1); ln = rng.randrange(4, 600)
                for _ in range(ln): buf.append(buf[-dist])
            else: buf += synth.gen_text_zipf(rng.randrange(1000), rng.randrange(1, 500)).tobytes()
        inputs.append(bytes(buf[:n]))
    expect(inputs, run(inputs))


def test_ring_skips_and_far_candidates():
    # matches longer than the feeder's read-ahead (the feeder skips what the match jumped over), candidates exactly 65535 and 65536
    # back (:201), a cursor deep inside a prefix (the ring starts 64 KiB before it), a literal run longer than the ring
    t = synth.gen_text_zipf(21, 3000).tobytes()
    r = vectors.rng_bytes(9, 400000)
    a = vectors.rng_bytes(1, 40)
    inputs = [t + bytes(200000) + t + bytes(70000) + t * 2,
              t * 90,                                                   # distance 3000 for 270 000 bytes: one long match, then text
              a + vectors.rng_bytes(2, 65535 - 40) + a + vectors.rng_bytes(3, 65536 - 40) + a + t,     # distances 65535 (legal) and 65536 (not)
              r[:300000] + t + r[300000:310000] + t,                  # a literal run of 300 000 (emitter: HBM), then matches at distance ~13 000
              t + r[:140000] + t[:200] + bytes(150000) + t[:300]]
    expect(inputs, run(inputs))
    data = synth.silesia_mix(8 << 20, (8 << 20) + 400000).tobytes()
    inputs = [data, data[:300000], data]
    cursors = [250000, 131072, 66000]
    expect(inputs, run(inputs, cursors=cursors), cursors=cursors)


def test_caller_owned_tables_with_offsets_linked_blocks():
    """Round 6: with the general kernel behind it (alone = 0) the team kernel takes caller-owned U32 tables at any EncoderTable::offset
    (mod.rs:30,:65-74) — the linked-blocks loop of framed/compress.rs:222-276: in_buffer = window ++ block, compress2 from the window's
    end, table.offset(what the window forgets).  After every block the output AND the table's 4096 slots + offset equal the oracle's;
    a refused block (cap = N on noise) is handed back (internal status, nothing written back: the general kernel's job on the device)."""
    WINDOW = 65536
    rng = random.Random(11)
    streams = [synth.silesia_mix(3 << 20, (3 << 20) + 400_000).tobytes(),
               synth.repeat256(150_000).tobytes() + synth.gen_random(5, 130_000).tobytes() + synth.silesia_mix(0, 120_000).tobytes(),
               b"".join(bytes([rng.randrange(4)]) * rng.randrange(1, 40) for _ in range(9000))]
    refused = compared_behind_offset = 0
    for data, bs in zip(streams, (65536, 40_000, 65536)):
        dic = data[:5000]
        t_emu, t_or = o.new_table(), o.new_table()
        # (the template of framed/compress.rs:202-211 is not the subject here: both sides start from the same seeded table)
        buf = dic
        o.compress2(dic + b"\0" * 16, cursor=len(dic) + 16, table=t_or); C.memmove(C.addressof(t_emu), C.addressof(t_or), C.sizeof(t_or))
        pos = len(dic)
        k = 0
        while pos < len(data):
            block = data[pos:pos + bs]; pos += len(block)
            inb = buf + block
            cap = len(block) if k % 3 else None               # framed/compress.rs:242 (cap = N) on most blocks
            erc, eout = o.compress2(inb, cursor=len(buf), table=t_or, cap=cap)
            before = bytes(t_emu)
            (rc, out), = emu_ffi.compress_batch([inb], cursors=[len(buf)], caps=[cap], tables=[t_emu], writable=True, alone=0)[0]
            if erc == 5:                                       # LZF_OUTPUT_FULL: handed back, the caller's table untouched
                assert rc == -30000 and bytes(t_emu) == before, (k, rc)
                refused += 1
                C.memmove(C.addressof(t_emu), C.addressof(t_or), C.sizeof(t_or))      # (what the general kernel leaves: tests/test_gpu_parity.py)
            else:
                assert (rc, out) == (erc, eout), (k, rc, erc, len(out), len(eout))
                assert bytes(t_emu) == bytes(t_or), (k, "table after the block")
                compared_behind_offset += t_or.offset > 0
            buf = inb
            if len(buf) > WINDOW:                              # framed/compress.rs:271-275
                forget = len(buf) - WINDOW
                for t in (t_emu, t_or):
                    t.offset += forget
                buf = buf[forget:]
            k += 1
        assert t_or.offset > 0 and k >= 3
    assert refused >= 1 and compared_behind_offset >= 6, (refused, compared_behind_offset)
