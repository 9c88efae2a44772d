"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs — bit-exact for compress2 output bytes, decoded bytes and error kinds."""
import json
import os

import numpy as np
import pytest

import oracle_ffi as o
import vectors
import rust_lz_fear_amd  # noqa: F401
from rust_lz_fear_amd import ffi, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gpu_compress(items):
    return ffi.compress_blocks_host(items)


def gpu_decompress(items):
    return ffi.decompress_blocks_host(items)


def test_device_present():
    assert ffi.device_count() >= 1


def test_plain_c_consumer_of_the_abi(tmp_path):
    """examples/c_abi_roundtrip.c: the boundary from C with gcc — no Python, torch or C++ in the caller."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_roundtrip")
    libdir = os.path.join(root, "rust-lz-fear_amd")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-Wall", os.path.join(root, "examples", "c_abi_roundtrip.c"),
                           "-I", os.path.join(root, "include"), "-L", libdir, "-llzfear_hip", "-Wl,-rpath," + libdir, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "c abi ok" in r.stdout


# ---------------------------------------------------------------- decompress
def test_decode_kats():
    res = gpu_decompress([dict(input=d) for d, _, _ in vectors.DECODE_KATS])
    for (d, status, expect), (rc, out) in zip(vectors.DECODE_KATS, res):
        assert rc == status, d
        if expect is not None:
            assert out == expect


def all_cases():
    return vectors.small_cases() + vectors.medium_cases() + [(f"librs{i}", s) for i, s in enumerate(vectors.LIB_RS_STRINGS)]


def test_decompress_valid_blocks_matches_original():
    cases = all_cases()
    comps = [o.compress2(d)[1] for _, d in cases]
    res = gpu_decompress([dict(input=c, limit=max(len(d), 1), out_cap=len(d) + len(c) + 64) for c, (_, d) in zip(comps, cases)])
    for (name, d), (rc, out) in zip(cases, res):
        assert rc == 0, name
        assert out == d, name


def test_decompress_u16_streams_and_c_streams():
    cases = [(n, d) for n, d in all_cases() if 0 < len(d) <= 0xFFFF]
    comps = [o.compress2(d, kind=o.TABLE_U16)[1] for _, d in cases]
    res = gpu_decompress([dict(input=c, limit=len(d)) for c, (_, d) in zip(comps, cases)])
    for (name, d), (rc, out) in zip(cases, res):
        assert (rc, out) == (0, d), name


def mutate(rng, comp):
    b = bytearray(comp)
    kind = rng.integers(0, 5)
    if kind == 0 and len(b) > 1:           # truncate
        del b[rng.integers(1, len(b)):]
    elif kind == 1 and len(b) > 0:         # flip a byte
        i = rng.integers(0, len(b)); b[i] ^= 1 << rng.integers(0, 8)
    elif kind == 2 and len(b) > 0:         # set a byte to 0 / 0xFF (zero offsets, LSIC runs)
        i = rng.integers(0, len(b)); b[i] = 0 if rng.integers(0, 2) else 0xFF
    elif kind == 3:                        # append garbage
        b += bytes(rng.integers(0, 256, rng.integers(1, 9), dtype=np.uint8))
    else:                                  # several flips
        for _ in range(4):
            if len(b):
                i = rng.integers(0, len(b)); b[i] = rng.integers(0, 256)
    return bytes(b)


def test_decompress_malformed_inputs_same_error_kind():
    """The decode corpus idea (fuzz/fuzz_targets/decode.rs) at block level: mutated blocks must
    give the same DecodeError kind as the reference restatement, and the same bytes when Ok."""
    rng = np.random.default_rng(12345)
    base = [(n, d) for n, d in all_cases() if 0 < len(d) <= 300000]
    items, expect = [], []
    for name, d in base:
        comp = o.compress2(d)[1]
        for k in range(6):
            m = mutate(rng, comp)
            limit = len(d) if k % 2 == 0 else len(d) // 2 + 1
            cap = limit + len(m) + 64
            erc, eout = o.decompress_raw(m, limit=limit, cap=cap)
            items.append(dict(input=m, limit=limit, out_cap=cap))
            expect.append((name, k, erc, eout))
    res = gpu_decompress(items)
    kinds = set()
    for (name, k, erc, eout), (rc, out) in zip(expect, res):
        assert rc == erc, (name, k, rc, erc)
        kinds.add(rc)
        if rc == 0:
            assert out == eout, (name, k)
    assert {0, 1, 2, 3, 4} <= kinds      # every DecodeError variant was exercised


def test_decompress_prefix_and_existing_output():
    d = synth.gen_text_zipf(31, 50000).tobytes()
    dic, payload = d[:20000], d[20000:]
    # compress payload with the dictionary as addressable prefix
    rc, comp = o.compress2(dic + payload, cursor=len(dic))
    assert rc == 0
    items = [
        dict(input=comp, prefix=dic, limit=len(payload)),                       # dictionary as prefix
        dict(input=comp, existing=dic, limit=len(d)),                           # dictionary as existing output
        dict(input=comp, prefix=dic[:100], limit=len(payload)),                 # truncated dictionary
        dict(input=comp, prefix=dic[:10000], existing=dic[10000:], limit=len(d)),  # split
    ]
    res = gpu_decompress(items)
    exp = [o.decompress_raw(comp, prefix=it.get("prefix", b""), existing=it.get("existing", b""), limit=it["limit"]) for it in items]
    for (rc, out), (erc, eout) in zip(res, exp):
        assert rc == erc
        if rc == 0:
            assert out == eout
    assert res[0] == (0, payload) and res[1] == (0, d) and res[3] == (0, dic[10000:] + payload)
    assert res[2][0] == o.INVALID_DEDUP_OFFSET


def test_decompress_overlap_offsets():
    """copy_overlapping arms (decompress.rs:100-135): offsets 1..40 with long matches."""
    items, exp = [], []
    for off in list(range(1, 41)) + [63, 64, 65, 255, 256, 257, 1000]:
        for mlen in (4, 5, 19, 70, 300, 5000):
            lit = bytes((i * 7 + off) & 0xFF for i in range(off))
            tok_l = min(len(lit), 15); tok_m = min(mlen - 4, 15)
            blk = bytes([(tok_l << 4) | tok_m])
            if len(lit) >= 15:
                v = len(lit) - 15
                blk += b"\xff" * (v // 255) + bytes([v % 255])
            blk += lit + off.to_bytes(2, "little")
            if mlen - 4 >= 15:
                v = mlen - 4 - 15
                blk += b"\xff" * (v // 255) + bytes([v % 255])
            items.append(dict(input=blk))
            exp.append(o.decompress_raw(blk))
    res = gpu_decompress(items)
    for e, r in zip(exp, res):
        assert e[0] == 0 and r == e


def _analysis_env(**kw):
    """The analysis flavour of the library (every kernel generation + the environment knobs that select them;
    rust-lz-fear_amd/build.py).  The product library has neither."""
    from rust_lz_fear_amd import build
    path = build.build_analysis_library()
    return dict(os.environ, LZF_LIB_PATH=path, **kw)


@pytest.mark.parametrize("variant", ["auto", "wave", "staged16", "staged32", "direct4w", "paired16", "paired24", "paired48", "paired256", "seg", "fed", "fed3", "ordered"])
def test_every_decompress_kernel_generation(variant):
    """Every kernel generation kept in the analysis library (and every ring/region geometry) implements the same contract.
    "ordered": the longest-first launch order that large batches get, forced on for these small ones.
    "seg": the segmented pipeline (one block decoded by many wavefronts) with its size window opened to every input — small
    blocks, handcrafted streams; malformed, prefix and existing-output jobs take its hand-over to the pair kernel.
    "fed" / "fed3": the bitmap-fed kernel of large batches (round 6) forced for these small ones, whole jobs / every job in three pieces."""
    import subprocess, sys
    env = _analysis_env(LZF_DECOMPRESS_KERNEL=variant) if variant != "ordered" else _analysis_env(LZF_DECOMPRESS_ORDER="always")
    if variant.startswith("fed"):       # the bitmap-fed kernel for every batch size and input size; "fed3": every job handed on twice (three pieces)
        env = _analysis_env(LZF_DECOMPRESS_KERNEL="fed", LZF_FED_MIN_IN="1", **({"LZF_FED_PIECES": "3"} if variant == "fed3" else {}))
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "variant_check.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "variant ok" in r.stdout


@pytest.mark.parametrize("kernel", ["team", "compact", "general", "ordered"])
def test_every_compress_kernel(kernel):
    """Fresh-table U32 jobs through the team kernel of the latency class (the default for a batch this small: three wavefronts per
    block, input ring and table in LDS — the same source the CPU suite runs under the lock-step emulator,
    tests/test_emu_compress_team.py), the compact-table kernel (the default beyond one block per CU) and the general kernel: same
    bytes as the oracle, over inputs that cross several 64 KiB epochs / ring wraps, skip epochs inside one match and widen the batches.
    "ordered": the cost probe + longest-first queue order that large batches get, forced on for this small one."""
    import subprocess, sys
    env = _analysis_env(LZF_COMPRESS_KERNEL=kernel) if kernel != "ordered" else _analysis_env(LZF_COMPRESS_ORDER="always")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "compress_variant_check.py")], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "variant ok" in r.stdout


def test_product_dispatch_every_batch_size_class():
    """The product library has no knobs: the batch size alone picks the kernel (decompress: the segmented pipeline up to 4 blocks per CU,
    paired48 up to 8, the bitmap-fed kernel beyond — paired24 / staged16 when it declines a call; compress: the team kernel up to one block per CU, the compact kernel beyond) and the launch
    order (longest first beyond 8 blocks per CU for decompress, beyond 18 per CU for compress).  Batches on both sides of every threshold, small blocks so the oracle keeps up."""
    rng = np.random.default_rng(5)
    base = synth.silesia_mix(0, 1 << 20)
    def blocks(n):
        out = []
        for i in range(n):
            a = int(rng.integers(0, (1 << 20) - 3000)); ln = int(rng.integers(1, 2500))
            out.append(base[a:a + ln].tobytes())
        return out
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for n in (cus - 6, cus, cus + 44, 2100, 4700, 16500):
        bl = blocks(n)
        comp = gpu_compress([dict(input=b, out_cap=len(b) + len(b) // 200 + 32) for b in bl])
        # the compressor's latency class: no more jobs than compute units -> a team of wavefronts and a CU's LDS per block
        launch = ffi.lib().lzf_last_compress_launch().decode()
        assert launch.startswith("lzf_compress_team_kernel" if n <= cus else "lzf_compress_compact_kernel"), (n, launch)
        step = max(1, n // 400)
        for b, (rc, c) in list(zip(bl, comp))[::step]:
            erc, ec = o.compress2(b, cap=len(b) + len(b) // 200 + 32)
            assert rc == erc and (rc != 0 or c == ec)
        ok = [(b, c) for b, (rc, c) in zip(bl, comp) if rc == 0]
        dec = gpu_decompress([dict(input=c, limit=len(b), out_cap=len(b) + len(c) + 64) for b, c in ok])
        for (b, c), (rc, d) in zip(ok, dec):
            assert rc == 0 and d == b


def _seg_mixed_items(rng):
    """Blocks for the segmented pipeline (compressed size >= 64 KiB puts a job in its window) next to jobs it must leave to
    the pair kernel.  Returns (items for gpu_decompress, expected (status, bytes) from the oracle)."""
    mib = 1 << 20
    gens = [synth.gen_text_zipf, synth.gen_markup, synth.gen_exe, synth.gen_records, synth.gen_walk16, synth.gen_log]
    raws = [g(100 + i, mib + 777 * i).tobytes() for i, g in enumerate(gens)]
    raws.append(synth.silesia_mix(0, 3 * mib).tobytes())                                  # a 3 MiB text block
    raws.append(bytes(2 * mib) + synth.gen_text_zipf(7, 300000).tobytes() + bytes(mib))     # zero runs of MiBs: sequences larger than any sub-batch
    raws.append((synth.repeat256(700000).tobytes() + synth.gen_random(3, 200000).tobytes()) * 2)   # 64 KiB+ overlapping matches, 200 KB literal runs
    raws.append(synth.gen_random(9, 300000).tobytes() + synth.gen_text_zipf(8, 400000).tobytes())  # one 300 KB literal run, then text
    items, exp = [], []
    for d in raws:
        c = o.compress2(d)[1]
        items.append(dict(input=c, limit=len(d), out_cap=len(d) + len(c) + 64)); exp.append((0, d))
    # handcrafted streams (dense tokens, long overlapping matches, tiny offsets), large enough for the window
    for seed, nseq, prof in [(21, 60000, "dense"), (22, 30000, "mixed"), (23, 3000, "long"), (24, 20000, "rle")]:
        blk, out = vectors.synth_stream(seed, nseq, prof)
        items.append(dict(input=blk, limit=len(out), out_cap=len(out) + len(blk) + 64)); exp.append((0, out))
    d0 = raws[0]; c0 = o.compress2(d0)[1]
    # the same block with too little room, with a limit one byte short, and damaged: pair-kernel statuses
    for kw in (dict(limit=len(d0), out_cap=len(d0) - 5000), dict(limit=len(d0) - 1, out_cap=len(d0) + len(c0) + 64)):
        items.append(dict(input=c0, **kw)); exp.append(o.decompress_raw(c0, limit=kw["limit"], cap=kw["out_cap"]))
    for k in range(6):
        b = bytearray(c0)
        for _ in range(1 + k):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        if k >= 4:
            del b[int(rng.integers(len(b) // 2, len(b))):]
        m = bytes(b)
        items.append(dict(input=m, limit=len(d0), out_cap=len(d0) + len(m) + 64)); exp.append(o.decompress_raw(m, limit=len(d0), cap=len(d0) + len(m) + 64))
    # prefix / existing output (not the pipeline's), and a block below its window
    dic, payload = d0[:60000], d0[60000:]
    cp = o.compress2(dic + payload, cursor=len(dic))[1]
    items.append(dict(input=cp, prefix=dic, limit=len(payload))); exp.append((0, payload))
    items.append(dict(input=cp, existing=dic, limit=len(d0))); exp.append((0, d0))
    small = d0[:50000]
    items.append(dict(input=o.compress2(small)[1], limit=len(small))); exp.append((0, small))
    return items, exp


def test_segmented_pipeline_mixed_batch():
    """lzf_decompress_batch sends batches of up to four blocks per CU through the segmented pipeline (a block decoded by
    many wavefronts: speculative chunk parse, seam, tile scan, records + literals, levels, stager / resolver pair) and hands
    what the pipeline does not finish — every DecodeError, capacity, prefix / existing output, blocks below 64 KiB of input —
    to the pair kernel.  One batch with all of it: statuses and bytes as the oracle's."""
    rng = np.random.default_rng(77)
    items, exp = _seg_mixed_items(rng)
    res = gpu_decompress(items)
    nerr = 0
    for i, ((rc, out), (erc, eout)) in enumerate(zip(res, exp)):
        assert rc == erc, (i, rc, erc)
        if rc == 0:
            assert out == eout, i
        else:
            nerr += 1
    assert nerr >= 2


def test_segmented_pipeline_grouped_mixed_batch():
    """Calls of 32 jobs and more take the pipeline's last two stages in groups (capi.hip: seg_groups — the records stage of the next
    quarter of the jobs, by sequences, under the resolve stage of the one before, on streams of the library's own).  The mixed
    batch twice over, so that every group holds jobs the pipeline finishes, jobs it fails (damaged blocks, capacity, limit) and jobs
    it never takes (prefix, existing output, below the window): statuses and bytes as the oracle's, whatever group a job fell into."""
    rng = np.random.default_rng(78)
    items, exp = _seg_mixed_items(rng)
    items, exp = items * 2, exp * 2
    assert len(items) >= 32
    res = gpu_decompress(items)
    for i, ((rc, out), (erc, eout)) in enumerate(zip(res, exp)):
        assert rc == erc, (i, rc, erc)
        if rc == 0:
            assert out == eout, i


def test_segmented_pipeline_grouped_beside_application_streams():
    """The grouped call on a stream of the application's while other application streams are busy (HIP runs a handful of hardware
    queues: the library's resolve streams must not depend on having one to themselves for correctness), twice in a row without a
    synchronisation in between (the second call's scratch and events reuse the first's)."""
    import torch
    from rust_lz_fear_amd import device
    mib = 1 << 20
    base = [synth.silesia_mix(k * 4 * mib, k * 4 * mib + mib).tobytes() for k in (0, 17, 29, 47)] + [synth.gen_records(5, mib).tobytes(), synth.gen_log(7, mib).tobytes()]
    pairs = [(d, c) for d, (rc, c) in ((d, o.compress2(d)) for d in base) if rc == 0]
    n = len(pairs) * 16
    assert 32 <= n <= 256
    blob = b"".join(c for _, c in pairs)
    d_in = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    offs = np.cumsum([0] + [len(c) for _, c in pairs])
    d_out = torch.zeros(n * mib, dtype=torch.uint8, device="cuda")
    dj = np.zeros(n, dtype=device.DJOB)
    for k in range(n):
        i = k % len(pairs)
        dj["input"][k] = d_in.data_ptr() + int(offs[i]); dj["input_len"][k] = len(pairs[i][1])
        dj["out"][k] = d_out.data_ptr() + k * mib; dj["out_cap"][k] = mib; dj["output_limit"][k] = mib
    d_dj = device.to_device(dj, "cuda")
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
    mine, busy1, busy2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.randn(2048, 2048, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(busy1):
        for _ in range(20): a1 = a @ a
    with torch.cuda.stream(busy2):
        for _ in range(20): a2 = a @ a
    with torch.cuda.stream(mine):
        device.decompress_batch(d_dj, d_res, n)
        d_out.zero_(); d_res.zero_()
        device.decompress_batch(d_dj, d_res, n)
    torch.cuda.synchronize()
    assert "segmented" in ffi.lib().lzf_last_decompress_launch().decode()
    r = device.results_to_host(d_res, n)
    for k in range(n):
        d = pairs[k % len(pairs)][0]
        assert r["status"][k] == ffi.OK and int(r["out_len"][k]) == len(d), k
        assert bytes(d_out[k * mib:k * mib + len(d)].cpu().numpy()) == d, k


@pytest.mark.parametrize("seed,nseq", [(41, 30000), (42, 60000), (43, 120000)])
def test_segmented_pipeline_every_copy_class(seed, nseq):
    """Handcrafted streams that hit every copy path of the resolver (lz4_decompress_seg.hip): the four two-ended sizes with and
    without run-length offsets (1, 2, 4 — classes 1-4 and 9-12), long run-length matches stored by the whole wave, overlapping
    ones, 65..160-byte matches alone and four or more to a level, matches of thousands of bytes — at every ring position (the
    streams decode to several MiB).  The oracle agrees with the generator; the GPU with both."""
    blk, out = vectors.synth_stream(seed, nseq, "classes")
    assert len(blk) >= 65536                                          # inside the pipeline's window
    erc, eout = o.decompress_raw(blk, limit=len(out), cap=len(out) + len(blk) + 64)
    assert erc == 0 and eout == out
    items = [dict(input=blk, limit=len(out), out_cap=len(out) + len(blk) + 64)] * 3     # (three jobs: different output alignments)
    for rc, got in gpu_decompress(items):
        assert rc == 0 and got == out
    assert ffi.lib().lzf_last_decompress_launch().decode().startswith("segmented")


@pytest.mark.parametrize("copies", [25, 55])
def test_segmented_pipeline_smaller_rings(copies):
    """More than one block per CU: 64 KiB (up to two per CU) and 32 KiB rings (beyond), where the oldest sources are read
    back from HBM by the stager.  The 1 MiB blocks of a text / records / binary mix, `copies` times."""
    mib = 1 << 20
    base = [synth.silesia_mix(k * 4 * mib, k * 4 * mib + mib).tobytes() for k in (0, 3, 15, 17, 25, 29, 37, 44, 47)] + \
           [synth.gen_records(5, mib).tobytes(), synth.gen_exe(6, mib).tobytes(), synth.gen_log(7, mib).tobytes()]
    comps = [o.compress2(d) for d in base]
    pairs = [(d, c) for d, (rc, c) in zip(base, comps) if rc == 0]
    items = [dict(input=c, limit=len(d), out_cap=len(d) + len(c) + 64) for d, c in pairs] * copies
    assert 256 < len(items) <= 1024
    res = gpu_decompress(items)
    for k, (rc, out) in enumerate(res):
        assert rc == 0 and out == pairs[k % len(pairs)][0], k


def test_oversized_but_ok_literals_and_out_capacity():
    """raw/decompress.rs:63-67: literals are appended without looking at output_limit; only a match is checked
    (:96-98).  A block that ends in literals past the limit is Ok and longer than the limit — given room:
    with out_cap below what the reference's Vec would have grown to, the job reports LZF_OUT_CAPACITY instead."""
    lit = bytes(range(200))
    blk = bytes([0xF0, 200 - 15]) + lit                                  # one sequence: 200 literals, end of block
    m = bytes([0x40]) + b"abcd" + (4).to_bytes(2, "little")             # 4 literals + a match of 4 at offset 4
    cases = [dict(input=blk, limit=10, out_cap=400), dict(input=blk, limit=0, out_cap=200), dict(input=blk, limit=10, out_cap=199),
             dict(input=m + blk, limit=8, out_cap=400), dict(input=m + blk, limit=7, out_cap=400), dict(input=m + blk, limit=8, out_cap=100)]
    res = gpu_decompress(cases)
    exp = [o.decompress_raw(c["input"], limit=c["limit"], cap=c["out_cap"]) for c in cases]
    assert res[0] == (ffi.OK, lit) and res[1] == (ffi.OK, lit)
    assert res[2][0] == ffi.OUT_CAPACITY
    assert res[3] == (ffi.OK, b"abcdabcd" + lit)
    assert res[4][0] == ffi.MEMORY_LIMIT_EXCEEDED
    assert res[5][0] == ffi.OUT_CAPACITY
    for (rc, out), (erc, eout) in zip(res, exp):
        assert rc == erc and (rc != 0 or out == eout)


def test_compress2_for_any_writer_replays_the_reference_write_calls():
    """lzf_compress2_host_writer = compress2<W: Write, T>(input, cursor, table, writer) (mod.rs:165-166) without a capacity
    argument: the device compresses against the worst-case bound, then the reference's write calls are replayed into the
    writer (token, length tail 4 + 1 bytes at a time, literals, offset, length tail: mod.rs:150-163, :243-260).  A sink that
    refuses the first call that does not fit a budget (NoPartialWrites, framed/compress.rs:294-314) — at EVERY budget from 0
    to the stream's length: the accepted bytes, the status and the table afterwards equal the oracle's compress2 into the
    same sink; the refusing writer's error code comes back."""
    import ctypes as C
    L = ffi.lib()
    WRITE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
    L.lzf_compress2_host_writer.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, WRITE, C.c_void_p, C.POINTER(C.c_int)]
    text = synth.gen_text_zipf(4, 900).tobytes()
    data = text[:300] + bytes(40) + text[100:420] + bytes([7]) * 700 + text[:64]     # literals, zero runs, a 700-byte match (length tail of 3 bytes)
    for kind in (ffi.TABLE_U32, ffi.TABLE_U16):
        for cursor in (0, 100):
            full = o.compress2(data, cursor=cursor, kind=kind)[1]
            assert 20 < len(full) < 600
            for budget in list(range(0, len(full) + 2)):
                got = bytearray(); calls = []
                def sink(ctx, p, n):
                    calls.append(n)
                    if len(got) + n > budget:
                        return 42
                    got.extend(bytes(p[:n]))
                    return 0
                t = ffi.U32Table() if kind == ffi.TABLE_U32 else ffi.U16Table()
                te = o.new_table(kind)
                werr = C.c_int(0)
                rc = L.lzf_compress2_host_writer(data, len(data), cursor, C.addressof(t), kind, WRITE(sink), None, C.byref(werr))
                erc, eout = o.compress2(data, cursor=cursor, kind=kind, table=te, cap=budget)
                assert rc == erc, (kind, cursor, budget, rc, erc)
                assert bytes(got) == eout, (kind, cursor, budget)
                assert bytes(t) == bytes(te), (kind, cursor, budget)
                assert werr.value == (42 if rc == ffi.OUTPUT_FULL else 0)
                assert all(n in (1, 2, 4) or n > 0 for n in calls)
    # the empty payload writes nothing (mod.rs:171), a NULL table is T::default()
    calls = []
    rc = L.lzf_compress2_host_writer(data, len(data), len(data), None, ffi.TABLE_U32, WRITE(lambda c, p, n: calls.append(n) or 0), None, None)
    assert rc == 0 and calls == []


def test_compress_cursor_past_the_end_is_ok_and_empty():
    """compress2 with cursor >= input.len(): the loop at mod.rs:171 never runs -> Ok(()), nothing written, table untouched."""
    data = synth.text_zipf_64k().tobytes()[:5000]
    t = ffi.U32Table()
    res = gpu_compress([dict(input=data, cursor=len(data), out_cap=100), dict(input=data, cursor=len(data) + 7, out_cap=100, table=t),
                        dict(input=data, cursor=len(data) + 7, out_cap=100, kind=ffi.TABLE_U16)])
    assert res == [(0, b""), (0, b""), (0, b"")]
    assert not any(t.dict) and t.offset == 0
    assert o.compress2(data, cursor=len(data) + 7, cap=100) == (0, b"")


def test_decompress_handcrafted_streams():
    """Blocks assembled sequence by sequence (tests/vectors.py synth_stream): shapes a greedy
    compressor never emits — > 1024 tokens per 8 KiB (token-list cut), multi-KiB literal runs and
    matches (solo path, cooperative paths), 90 000-byte overlapping matches, offsets 1..8 and
    near 65 535 — must decode like the reference restatement, at every output-limit outcome."""
    cases = []
    for seed, n, prof in [(1, 6000, "dense"), (2, 3000, "mixed"), (3, 400, "long"), (4, 2000, "rle"),
                          (5, 20000, "mixed"), (6, 150, "long"), (7, 9000, "dense"), (8, 5000, "rle")]:
        cases.append(vectors.synth_stream(seed, n, prof))
    items = []
    for blk, out in cases:
        items.append(dict(input=blk, limit=len(out), out_cap=len(out) + len(blk) + 64))
        items.append(dict(input=blk, limit=len(out) // 2, out_cap=len(out) + len(blk) + 64))     # MemoryLimitExceeded somewhere
        items.append(dict(input=blk[: len(blk) * 2 // 3], limit=len(out), out_cap=len(out) + len(blk) + 64))   # truncated
    res = gpu_decompress(items)
    for it, (rc, got) in zip(items, res):
        erc, eout = o.decompress_raw(it["input"], limit=it["limit"], cap=it["out_cap"])
        assert rc == erc
        if rc == 0:
            assert got == eout
    for (blk, out), (rc, got) in zip(cases, res[::3]):
        assert rc == 0 and got == out


# ---------------------------------------------------------------- compress
def test_compress_u32_bit_exact():
    cases = all_cases()
    res = gpu_compress([dict(input=d) for _, d in cases])
    for (name, d), (rc, comp) in zip(cases, res):
        erc, ecomp = o.compress2(d)
        assert rc == erc == 0, name
        assert comp == ecomp, (name, len(comp), len(ecomp))


def test_compress_u16_bit_exact():
    cases = [(n, d) for n, d in all_cases() if len(d) <= 0xFFFF]
    res = gpu_compress([dict(input=d, kind=ffi.TABLE_U16) for _, d in cases])
    for (name, d), (rc, comp) in zip(cases, res):
        erc, ecomp = o.compress2(d, kind=o.TABLE_U16)
        assert rc == erc == 0, name
        assert comp == ecomp, name
    rc, _ = gpu_compress([dict(input=bytes(65536), kind=ffi.TABLE_U16)])[0]
    assert rc == ffi.CONTRACT                                     # mod.rs:167


def test_compress_survey_fingerprints():
    S = json.load(open(os.path.join(GOLD, "survey_fingerprints.json")))
    e = bytearray(synth.lcg_bytes(1, 156)); e[143:149] = e[11:17]
    g1 = vectors.big_compression_bytes(10 ** 6)
    g3 = synth.lcg_bytes(5, 262144, 3)
    g4 = synth.lcg_bytes(7, 65535, 1)
    res = gpu_compress([dict(input=bytes(e)), dict(input=g1), dict(input=bytes(65536)), dict(input=g3),
                        dict(input=g4, kind=ffi.TABLE_U16), dict(input=g4)])
    fp = lambda b: [len(b), "%08x" % o.xxh32(b)]
    assert [rc for rc, _ in res] == [0] * 6
    assert fp(res[0][1]) == S["KAT-A"]["u32_raw"]                  # quirk B2
    assert fp(res[1][1]) == S["G1"]["u32_raw"]
    assert fp(res[2][1]) == S["G2"]["u32_raw"]
    assert fp(res[3][1]) == S["G3"]["u32_raw"]
    assert fp(res[4][1]) == S["G4"]["u16_raw"]
    assert fp(res[5][1]) == S["G4"]["u32_raw"]


def test_compress_output_full_cap():
    """NoPartialWrites cap = N (framed/compress.rs:242): incompressible blocks abort, C == N passes."""
    d = vectors.rng_bytes(11, 5000)
    rc, full = o.compress2(d)
    res = gpu_compress([dict(input=d, out_cap=len(d)), dict(input=d, out_cap=len(full)), dict(input=d, out_cap=len(full) - 1),
                        dict(input=d, out_cap=0), dict(input=b"", out_cap=0)])
    assert res[0][0] == ffi.OUTPUT_FULL
    assert res[1] == (0, full)
    assert res[2][0] == ffi.OUTPUT_FULL
    assert res[3][0] == ffi.OUTPUT_FULL
    assert res[4] == (0, b"")                                      # quirk B4


def test_compress_with_prefix_and_table_carry_linked_blocks():
    """cursor > 0 + carried table == the linked-blocks driver of framed/compress.rs:221-276."""
    data = synth.silesia_mix(20 << 20, (20 << 20) + 300000).tobytes()
    bs = 65536
    tg, to = ffi.U32Table(), o.new_table()
    buf = b""
    for off in range(0, len(data), bs):
        blk = data[off:off + bs]
        inp = buf + blk
        erc, ecomp = o.compress2(inp, cursor=len(buf), table=to, cap=len(blk))
        (rc, comp), = gpu_compress([dict(input=inp, cursor=len(buf), table=tg, out_cap=len(blk))])
        assert rc == erc, off
        if rc == 0:
            assert comp == ecomp, off
        assert bytes(tg) == bytes(to), off        # table state after the block is identical
        buf = inp
        if len(buf) > 65536:
            forget = len(buf) - 65536
            tg.offset += forget; to.offset += forget
            buf = buf[forget:]


def test_linked_streams_of_one_call_take_the_team_class_tables_exact():
    """Round 6 (review item 7): a linked-blocks call of no more streams than compute units — block k of 200 streams in launch k — runs in
    the compressor's latency class, lzf_compress_team_kernel, with the caller-owned tables carried (offset applied on the way in and out of
    LDS); the general kernel behind it takes what the team kernel hands back (a refused block: cap = N on noise) and U16 / contract jobs.
    Output bytes, statuses and the tables' bytes after EVERY block == the oracle's linked loop (framed/compress.rs:221-276)."""
    S, bs, nblk = 200, 65536, 4
    rng = np.random.default_rng(12)
    datas = []
    for k in range(S):
        a = int(rng.integers(0, 180 << 20))
        d = bytearray(synth.silesia_mix(a, a + bs * nblk - 1000 * (k % 7)).tobytes())
        if k % 9 == 4:
            d[bs + 100:2 * bs + 5000] = synth.gen_random(k, bs + 4900).tobytes()      # blocks 1 (and part of 2) of these streams are noise: refused at cap = N
        datas.append(bytes(d))
    tg = [ffi.U32Table() for _ in range(S)]
    to = [o.new_table() for _ in range(S)]
    bufs = [b""] * S
    refused = 0
    for b in range(nblk):
        items, exp = [], []
        for k in range(S):
            blk = datas[k][b * bs:(b + 1) * bs]
            inp = bufs[k] + blk
            exp.append(o.compress2(inp, cursor=len(bufs[k]), table=to[k], cap=len(blk)))
            items.append(dict(input=inp, cursor=len(bufs[k]), table=tg[k], out_cap=len(blk)))
        res = gpu_compress(items)
        assert ffi.lib().lzf_last_compress_launch().decode() == "lzf_compress_team_kernel + lzf_compress_team_carry_kernel + lzf_compress_wave_kernel"
        for k in range(S):
            assert res[k][0] == exp[k][0], (b, k, res[k][0], exp[k][0])
            if exp[k][0] == 0:
                assert res[k][1] == exp[k][1], (b, k)
            else:
                refused += 1
            assert bytes(tg[k]) == bytes(to[k]), (b, k, "table after the block")
            bufs[k] = items[k]["input"]
            if len(bufs[k]) > 65536:
                forget = len(bufs[k]) - 65536
                tg[k].offset += forget; to[k].offset += forget
                bufs[k] = bufs[k][forget:]
    assert refused >= 20 and all(t.offset > 0 for t in to)


def test_compress_dictionary_template_table():
    """Template table seeded from a dictionary (framed/compress.rs:202-214), read-only clone per block."""
    dic = synth.gen_text_zipf(3, 70000).tobytes()
    blocks = [synth.gen_text_zipf(40 + i, 65536).tobytes() for i in range(4)]
    # oracle: seed by the reference loop
    tmpl = o.new_table()
    import ctypes as C
    contract = C.c_int(0)
    rep = o.lib().lzfo_u32_replace
    rep.restype = C.c_size_t
    rep.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
    for off in range(0, len(dic) - 7, 3):
        rep(C.addressof(tmpl), dic, len(dic), off, C.byref(contract))
    items, exp = [], []
    for b in blocks:
        t = o.U32Table.from_buffer_copy(bytes(tmpl))
        exp.append(o.compress2(dic + b, cursor=len(dic), table=t, cap=len(b)))
        tg = ffi.U32Table.from_buffer_copy(bytes(tmpl))
        items.append(dict(input=dic + b, cursor=len(dic), table=tg, out_cap=len(b), readonly=True))
    res = gpu_compress(items)
    for e, r, it in zip(exp, res, items):
        assert r == e
        assert bytes(it["table"]) == bytes(tmpl)      # readonly: template untouched


def test_device_table_seeding_matches_reference_loop():
    import ctypes as C
    import torch
    dic = synth.gen_text_zipf(3, 70000).tobytes()
    tmpl = o.new_table()
    contract = C.c_int(0)
    rep = o.lib().lzfo_u32_replace
    rep.restype = C.c_size_t
    rep.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
    for off in range(0, len(dic) - 7, 3):
        rep(C.addressof(tmpl), dic, len(dic), off, C.byref(contract))
    d_dic = torch.frombuffer(bytearray(dic), dtype=torch.uint8).cuda()
    d_tab = torch.zeros(C.sizeof(ffi.U32Table), dtype=torch.uint8, device="cuda")
    ffi.check(ffi.lib().lzf_table_seed_from_dictionary(d_tab.data_ptr(), d_dic.data_ptr(), len(dic), None))
    torch.cuda.synchronize()
    assert d_tab.cpu().numpy().tobytes() == bytes(tmpl)
    for short in (0, 4, 7, 8, 9, 11):
        ffi.check(ffi.lib().lzf_table_seed_from_dictionary(d_tab.data_ptr(), d_dic.data_ptr(), short, None))
        t2 = o.new_table()
        for off in range(0, short - 7, 3):
            rep(C.addressof(t2), dic[:short], short, off, C.byref(contract))
        torch.cuda.synchronize()
        assert d_tab.cpu().numpy().tobytes() == bytes(t2), short


def test_xxh32_batch():
    import torch
    bufs = [vectors.rng_bytes(i, n) for i, n in enumerate([0, 1, 3, 4, 15, 16, 17, 31, 32, 33, 63, 64, 100, 1000, 65536, 300001] * 2 + [5] * 7)]
    blob = b"".join(bufs)
    d = torch.frombuffer(bytearray(blob + b"\0"), dtype=torch.uint8).cuda()
    offs = np.cumsum([0] + [len(b) for b in bufs[:-1]]).astype(np.uint64)
    ptrs = torch.from_numpy((offs + np.uint64(d.data_ptr())).view(np.int64)).cuda()
    lens = torch.from_numpy(np.array([len(b) for b in bufs], dtype=np.int64)).cuda()
    out = torch.zeros(len(bufs), dtype=torch.int32, device="cuda")
    ffi.check(ffi.lib().lzf_xxh32_batch(ptrs.data_ptr(), lens.data_ptr(), out.data_ptr(), len(bufs), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    for g, b in zip(got, bufs):
        assert int(g) == o.xxh32(b)


# ---------------------------------------------------------------- BASELINE sizes (4 MiB blocks)
def test_full_size_silesia_blocks_bit_exact_and_roundtrip():
    """configs[1]/[2]: every 4 MiB block of the 211 938 580-byte Silesia stand-in (51 blocks):
    GPU compress == oracle compress byte-for-byte (or both abort to 'stored'), GPU decompress of
    those bytes == the input.  Device-resident path (lzf_*_batch with HBM job arrays)."""
    import torch
    from rust_lz_fear_amd import device
    data = synth.silesia_mix()
    bs = 4 << 20
    d_in = torch.from_numpy(data).cuda()
    blocks = device.BlockSet(d_in, bs)
    assert blocks.n == 51
    d_out = torch.empty(blocks.n * bs, dtype=torch.uint8, device="cuda")
    jobs = blocks.compress_jobs(d_out, bs)
    d_jobs = device.to_device(jobs, "cuda")
    d_res = torch.zeros(blocks.n * 16, dtype=torch.uint8, device="cuda")
    device.compress_batch(d_jobs, d_res, blocks.n)
    torch.cuda.synchronize()
    res = device.results_to_host(d_res, blocks.n)
    comp_host = d_out.cpu().numpy()
    stored = 0
    dj = np.zeros(blocks.n, dtype=device.DJOB)
    d_dec = torch.zeros(blocks.n * bs, dtype=torch.uint8, device="cuda")
    for i in range(blocks.n):
        blk = data[i * bs:(i + 1) * bs].tobytes()
        erc, ecomp = o.compress2(blk, cap=len(blk))
        assert int(res["status"][i]) == erc, i
        if erc == 0:
            assert int(res["out_len"][i]) == len(ecomp), i
            assert comp_host[i * bs:i * bs + len(ecomp)].tobytes() == ecomp, i
        else:
            stored += 1
        dj["input"][i] = d_out.data_ptr() + i * bs
        dj["input_len"][i] = res["out_len"][i] if erc == 0 else 0
        dj["out"][i] = d_dec.data_ptr() + i * bs
        dj["out_cap"][i] = bs
        dj["output_limit"][i] = bs
    assert 0 < stored < 10          # the near-random segments (sao, x-ray) are stored raw
    d_dj = device.to_device(dj, "cuda")
    d_res2 = torch.zeros(blocks.n * 16, dtype=torch.uint8, device="cuda")
    device.decompress_batch(d_dj, d_res2, blocks.n)
    torch.cuda.synchronize()
    res2 = device.results_to_host(d_res2, blocks.n)
    dec = d_dec.cpu().numpy()
    for i in range(blocks.n):
        assert int(res2["status"][i]) == 0
        if int(res["status"][i]) == 0:
            n = int(blocks.lens[i])
            assert int(res2["out_len"][i]) == n
            assert np.array_equal(dec[i * bs:i * bs + n], data[i * bs:i * bs + n]), i


def test_copy_ranges_stored_blocks():
    """lzf_copy_ranges: arbitrary (unaligned, ragged, empty) device ranges copied in one launch."""
    import torch
    from rust_lz_fear_amd import device
    rng = np.random.default_rng(8)
    src = torch.from_numpy(rng.integers(0, 256, 3_000_000, dtype=np.uint8)).cuda()
    dst = torch.zeros(3_200_000, dtype=torch.uint8, device="cuda")
    lens = [0, 1, 15, 16, 17, 65535, 65536, 65537, 200_001, 1_000_003]
    so = [5, 77, 1001, 4096, 12345, 70_000, 140_001, 300_003, 700_007, 1_500_001]
    do, pos = [], 3
    for n in lens:
        do.append(pos); pos += n + 13
    as_dev = lambda a: torch.from_numpy(np.asarray(a, dtype=np.uint64).view(np.int64)).cuda()
    device.copy_ranges(as_dev([src.data_ptr() + x for x in so]), as_dev([dst.data_ptr() + x for x in do]), as_dev(lens),
                       len(lens), max(lens))
    torch.cuda.synchronize()
    exp = np.zeros(3_200_000, dtype=np.uint8); s = src.cpu().numpy()
    for n, a, b in zip(lens, so, do):
        exp[b:b + n] = s[a:a + n]
    assert np.array_equal(dst.cpu().numpy(), exp)


def test_blocks_from_another_encoder_hc_fixtures():
    """tests/golden/hc_blocks.*: blocks written by liblz4's LZ4_compress_HC (levels 9, 12) and LZ4_compress_fast — parse
    shapes the greedy lz-fear encoder never produces.  The GPU decodes the committed bytes to the regenerated inputs, at
    the exact limit, one byte under it (MemoryLimitExceeded like the oracle) and with a prefix cut off the front."""
    J = json.load(open(os.path.join(GOLD, "hc_blocks.json")))
    blob = open(os.path.join(GOLD, "hc_blocks.bin"), "rb").read()
    items, want = [], []
    for b in J["blocks"]:
        comp = blob[b["offset"]: b["offset"] + b["length"]]
        assert [len(comp), "%08x" % o.xxh32(comp)] == b["comp"]
        data = eval(b["input"], {"synth": synth}).tobytes()
        assert [len(data), "%08x" % o.xxh32(data)] == b["in"]
        items.append(dict(input=comp, limit=len(data), out_cap=len(data) + len(comp) + 64)); want.append((0, data))
        items.append(dict(input=comp, limit=len(data) - 1, out_cap=len(data) + len(comp) + 64)); want.append(o.decompress_raw(comp, limit=len(data) - 1, cap=len(data) + len(comp) + 64))
        items.append(dict(input=comp[: len(comp) // 2], limit=len(data), out_cap=len(data) + len(comp) + 64)); want.append(o.decompress_raw(comp[: len(comp) // 2], limit=len(data), cap=len(data) + len(comp) + 64))
    got = gpu_decompress(items)
    for (rc, out), (erc, eout), it in zip(got, want, items):
        assert rc == erc
        if rc == 0:
            assert out == eout
