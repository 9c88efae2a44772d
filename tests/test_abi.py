"""CPU tests of the drop-in boundary: the C-ABI library builds, loads, exports every symbol
include/lzfear_hip.h declares, agrees on struct layouts, and fails loudly without a device."""
import ctypes as C
import os
import re

import pytest

import rust_lz_fear_amd  # noqa: F401
from rust_lz_fear_amd import build, ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return ffi.lib()


def declared_functions(name="lzfear_hip.h"):
    hdr = open(os.path.join(ROOT, "include", name)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lzf_[a-z0-9_]+)\s*\(", hdr)))


def test_header_declares_expected_entry_points():
    names = declared_functions()
    assert "lzf_compress_batch" in names and "lzf_decompress_batch" in names
    assert sorted(names) == sorted(ffi.EXPORTS)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions() + declared_functions("lzfear_frame.h"):
        assert hasattr(lib, name), name
    assert lib.lzf_abi_version() == 2
    assert sorted(declared_functions("lzfear_frame.h")) == sorted(ffi.FRAME_EXPORTS)


def test_dist_library_exports_every_declared_symbol():
    """include/lzfear_dist.h (the block-sharded frame's exchange over RCCL) is what liblzfear_dist.so exports; the codec library
    itself does not depend on RCCL."""
    import subprocess
    from rust_lz_fear_amd import dist as lzdist
    path = build.build_dist_library()
    names = declared_functions("lzfear_dist.h")
    assert "lzf_frame_gather" in names and "lzf_dist_comm_init" in names
    L = lzdist.dist_lib()
    for name in names:
        assert hasattr(L, name), name
    needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
    assert "librccl" in needed and "liblzfear_hip" in needed
    core = subprocess.run(["readelf", "-d", build.build_library()], capture_output=True, text=True).stdout
    assert "rccl" not in core


def test_frame_layer_host_only_pieces(lib):
    """Header parsing, XXH32 and frame assembly are pure host code: usable without a GPU."""
    import ctypes as C
    import xxhash
    for n in (0, 1, 15, 16, 17, 1000):
        d = bytes(range(256)) * 4
        assert lib.lzf_xxh32(d[:n], n, 0) == xxhash.xxh32(d[:n]).intdigest()
    s = ffi.Settings()
    lib.lzf_settings_default(C.byref(s))
    assert (s.independent_blocks, s.block_checksums, s.content_checksum, s.block_size) == (1, 0, 1, 4 << 20)
    hc = (xxhash.xxh32(bytes([0x64, 0x70])).intdigest() >> 8) & 0xFF
    st = ffi.Xxh32State()
    lib.lzf_xxh32_reset(C.byref(st), 0)
    blob = bytes(range(256)) * 9
    for a, b in ((0, 5), (5, 16), (16, 1000), (1000, len(blob))):
        lib.lzf_xxh32_update(C.byref(st), blob[a:b], b - a)
    assert lib.lzf_xxh32_digest(C.byref(st)) == xxhash.xxh32(blob).intdigest()
    frame = bytes.fromhex("04224d186470") + bytes([hc]) + bytes(4) + (0x02CC5D05).to_bytes(4, "little")   # empty default frame
    info = ffi.FrameInfo()
    assert lib.lzf_frame_read_header(frame, len(frame), C.byref(info)) == 0
    assert info.block_maxsize == 4 << 20 and info.header_len == 7
    assert lib.lzf_frame_read_header(b"\x00" * 7, 7, C.byref(info)) == 17          # WrongMagic
    assert lib.lzf_frame_read_header(frame[:5], 5, C.byref(info)) == 16             # InputError


def test_table_replace_host_is_encoder_table_replace(lib):
    """lzf_table_replace_host / lzf_table_offset_host = EncoderTable::replace / ::offset (src/raw/compress/mod.rs:64-74, :88-99)
    on the C ABI's host tables: the same return values and the same table as the oracle's restatement, over a table whose
    offset is carried forward like a linked-block stream's (framed/compress.rs:271-275), the < 8-byte tail rule of
    hash_for_u32 (:43) and the contract violations (:67, :92).  Pure host code: no GPU needed."""
    import ctypes as C
    import numpy as np
    import oracle_ffi as o
    L = o.lib()
    L.lzfo_u32_replace.restype = C.c_size_t; L.lzfo_u16_replace.restype = C.c_size_t
    L.lzfo_u32_replace.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
    L.lzfo_u16_replace.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
    lib.lzf_table_replace_host.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.lzf_table_offset_host.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
    rng = np.random.default_rng(3)
    data = bytes(rng.integers(0, 7, 70000, dtype=np.uint8))              # few symbols: slots collide and repeat
    for kind, ours, ref, fn in ((ffi.TABLE_U32, ffi.U32Table(), o.U32Table(), L.lzfo_u32_replace), (ffi.TABLE_U16, ffi.U16Table(), o.U16Table(), L.lzfo_u16_replace)):
        n = len(data) if kind == ffi.TABLE_U32 else 30000
        for rnd in range(3):
            for pos in list(rng.integers(0, n - 8, 3000)) + [n - 8, n - 7, n - 5, n - 4]:
                pos = int(pos)
                prev = C.c_uint64(0); contract = C.c_int(0)
                rc = lib.lzf_table_replace_host(C.addressof(ours), kind, data, n, pos, C.byref(prev))
                exp = fn(C.addressof(ref), data, n, pos, C.byref(contract))
                assert (rc == ffi.CONTRACT) == bool(contract.value), (kind, pos, rc)
                if rc == 0:
                    assert prev.value == exp, (kind, pos)
            add = 4000 if kind == ffi.TABLE_U32 else 9000
            assert lib.lzf_table_offset_host(C.addressof(ours), kind, add) == 0
            ref.offset += add
            assert bytes(ours) == bytes(ref)
    t = ffi.U16Table()
    assert lib.lzf_table_replace_host(C.addressof(t), ffi.TABLE_U16, data, 100, 98, None) == ffi.CONTRACT      # fewer than 4 bytes (:59)
    t.offset = 65000
    assert lib.lzf_table_replace_host(C.addressof(t), ffi.TABLE_U16, data, 30000, 600, None) == ffi.CONTRACT   # beyond u16 (:92)
    t32 = ffi.U32Table(); t32.offset = 0xFFFFFFF0
    assert lib.lzf_table_replace_host(C.addressof(t32), ffi.TABLE_U32, data, 30000, 600, None) == ffi.CONTRACT  # beyond u32 (:67)
    assert lib.lzf_table_replace_host(C.addressof(t32), ffi.TABLE_U32, data, 100, 101, None) == ffi.CONTRACT


def test_struct_layouts_match_header():
    # sizes implied by the header (LP64): compress job 56 B, decompress job 64 B, result 16 B,
    # tables 16 KiB + 8 B
    assert C.sizeof(ffi.CompressJob) == 56
    assert C.sizeof(ffi.DecompressJob) == 64
    assert C.sizeof(ffi.JobResult) == 16
    assert C.sizeof(ffi.U32Table) == 4096 * 4 + 8 == C.sizeof(ffi.U16Table)
    from rust_lz_fear_amd import device
    assert device.CJOB.itemsize == 56 and device.DJOB.itemsize == 64 and device.RES.itemsize == 16
    assert [device.CJOB.fields[n][1] for n in ("input", "input_len", "cursor", "out", "out_cap", "table", "table_kind", "flags")] == \
        [ffi.CompressJob.input.offset, ffi.CompressJob.input_len.offset, ffi.CompressJob.cursor.offset,
         ffi.CompressJob.out.offset, ffi.CompressJob.out_cap.offset, ffi.CompressJob.table.offset,
         ffi.CompressJob.table_kind.offset, ffi.CompressJob.flags.offset]


def test_no_cpu_fallback(lib):
    """Without a HIP device every codec entry point must fail loudly — never compute on the CPU."""
    n = lib.lzf_device_count()
    if n > 0:
        pytest.skip("a GPU is present; the loud-failure path is for GPU-less hosts")
    assert n == ffi.E_NO_DEVICE
    with pytest.raises(ffi.LzfError):
        ffi.compress_blocks_host([dict(input=b"hello hello hello hello")])
    with pytest.raises(ffi.LzfError):
        ffi.decompress_blocks_host([dict(input=bytes([0x11, 97, 1, 0]))])
    assert b"no CPU fallback" in lib.lzf_last_error()


def test_product_does_not_reference_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may mention it."""
    bad = []
    for base in ("rust-lz-fear_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"oracle_ffi|lzf_oracle|liblzf_oracle|lzfo_", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
