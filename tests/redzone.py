"""Red-zone harness for the GPU codec (test infrastructure, -m gpu).

The reference is `#![forbid(unsafe_code)]` (src/lib.rs:1) and its fuzz target's only property is "no crash, no out-of-bounds on
arbitrary bytes" (fuzz/fuzz_targets/decode.rs), with the documented overshoot bound output_limit + input.len()
(src/raw/decompress.rs:55-57).  A GPU kernel has no such language guarantee, so it is checked from outside: every job's
buffers are carved out of larger device allocations

    input arena :  ... [input bytes][ >= 4 KiB of IN-poison ] ...          the bytes behind input_len are poison A in one run, B in another
    output arena:  ... [4 KiB OUT-poison][existing | out_cap bytes][4 KiB OUT-poison] ...

through the device-pointer entry points of the C ABI (lzf_decompress_batch / lzf_compress_batch).  After the call every
OUT-poison zone must be untouched — a write in front of `out` or behind `out + out_cap` is a failure even when it lands in slack
the plain tests never look at — and statuses and Ok bytes must be identical under both IN-poisons: a result that depends on bytes
behind `input_len` is an over-read that matters.
"""
import numpy as np
import torch

import rust_lz_fear_amd  # noqa: F401
from rust_lz_fear_amd import device, ffi

ZONE = 4096
OUT_POISON = 0xA5


def _layout(sizes, zone, rng, align_choices=(1, 2, 4, 8, 16, 64)):
    """Offsets of buffers of `sizes` bytes, each behind `zone` bytes of poison and at a start address with varying low bits."""
    offs, pos = [], 0
    for sz in sizes:
        pos += zone
        a = int(rng.choice(align_choices))
        pos = (pos + 63) // 64 * 64 + (a if a < 64 else 0) * int(rng.integers(0, 3))
        offs.append(pos)
        pos += sz
    return offs, pos + zone


def decompress_guarded(items, in_poison, seed=0, max_input_len=None):
    """items as ffi.decompress_blocks_host.  Returns (results [(status, bytes incl. existing)], zones_ok, detail)."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    n = len(items)
    ins = [bytes(it["input"]) for it in items]
    pres = [bytes(it.get("prefix", b"")) for it in items]
    exs = [bytes(it.get("existing", b"")) for it in items]
    limits = [it.get("limit") if it.get("limit") is not None else (1 << 63) - 1 for it in items]
    caps = []
    for it, ex, lim, d in zip(items, exs, limits, ins):
        cap = it.get("out_cap")
        caps.append(cap if cap is not None else len(ex) + min(lim, 1 << 26) + len(d) + 64)
    in_offs, in_total = _layout([len(d) for d in ins] + [len(p) for p in pres], ZONE, rng)
    out_offs, out_total = _layout(caps, ZONE, rng)
    h_in = np.full(in_total, in_poison, dtype=np.uint8)
    for o, d in zip(in_offs, ins + pres):
        h_in[o:o + len(d)] = np.frombuffer(d, dtype=np.uint8)
    h_out = np.full(out_total, OUT_POISON, dtype=np.uint8)
    for o, ex in zip(out_offs, exs):
        h_out[o:o + len(ex)] = np.frombuffer(ex, dtype=np.uint8)
    d_in = torch.from_numpy(h_in).to(dev)
    d_out = torch.from_numpy(h_out).to(dev)
    j = np.zeros(n, dtype=device.DJOB)
    j["input"] = np.uint64(d_in.data_ptr()) + np.array(in_offs[:n], dtype=np.uint64)
    j["input_len"] = [len(d) for d in ins]
    j["prefix"] = np.uint64(d_in.data_ptr()) + np.array(in_offs[n:], dtype=np.uint64)
    j["prefix_len"] = [len(p) for p in pres]
    j["out"] = np.uint64(d_out.data_ptr()) + np.array(out_offs, dtype=np.uint64)
    j["out_existing_len"] = [len(e) for e in exs]
    j["out_cap"] = caps
    j["output_limit"] = limits
    d_j = device.to_device(j, dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    device.decompress_batch(d_j, d_res, n, max_input_len=max_input_len)
    torch.cuda.synchronize()
    res = device.results_to_host(d_res, n)
    got = d_out.cpu().numpy()
    # every byte outside the jobs' [out, out + cap) must still be poison
    mask = np.ones(out_total, dtype=bool)
    for o, cap in zip(out_offs, caps):
        mask[o:o + cap] = False
    bad = np.nonzero(mask & (got != OUT_POISON))[0]
    detail = ""
    if len(bad):
        first = int(bad[0])
        k = int(np.searchsorted(np.array(out_offs), first, side="right")) - 1
        where = "behind" if k >= 0 and first >= out_offs[k] + caps[k] else "in front of"
        kk = k if where == "behind" else k + 1
        detail = f"{len(bad)} poisoned bytes overwritten; first at arena offset {first}: {where} job {kk}'s output slot"
    # the input arena is read-only for the codec
    if not np.array_equal(d_in.cpu().numpy(), h_in):
        detail += " | the INPUT arena was written to"
    out = []
    for i in range(n):
        ln = int(min(res["out_len"][i], caps[i]))
        out.append((int(res["status"][i]), got[out_offs[i]:out_offs[i] + ln].tobytes()))
    return out, detail == "", detail


def compress_guarded(items, in_poison, seed=0, kinds=None):
    """items as ffi.compress_blocks_host (input, cursor, kind, out_cap; fresh tables only).  Returns (results, zones_ok, detail)."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    n = len(items)
    ins = [bytes(it["input"]) for it in items]
    caps = [it["out_cap"] if it.get("out_cap") is not None else len(d) + len(d) // 255 + 64 for it, d in zip(items, ins)]
    in_offs, in_total = _layout([len(d) for d in ins], ZONE, rng)
    out_offs, out_total = _layout(caps, ZONE, rng)
    h_in = np.full(in_total, in_poison, dtype=np.uint8)
    for o, d in zip(in_offs, ins):
        h_in[o:o + len(d)] = np.frombuffer(d, dtype=np.uint8)
    d_in = torch.from_numpy(h_in).to(dev)
    d_out = torch.full((out_total,), OUT_POISON, dtype=torch.uint8, device=dev)
    j = np.zeros(n, dtype=device.CJOB)
    j["input"] = np.uint64(d_in.data_ptr()) + np.array(in_offs, dtype=np.uint64)
    j["input_len"] = [len(d) for d in ins]
    j["cursor"] = [it.get("cursor", 0) for it in items]
    j["out"] = np.uint64(d_out.data_ptr()) + np.array(out_offs, dtype=np.uint64)
    j["out_cap"] = caps
    j["table_kind"] = [it.get("kind", ffi.TABLE_U32) for it in items]
    if kinds is None:
        kinds = 0
        for it in items:
            kinds |= ffi.KINDS_U16 if it.get("kind", ffi.TABLE_U32) == ffi.TABLE_U16 else ffi.KINDS_U32
    d_j = device.to_device(j, dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    device.compress_batch(d_j, d_res, n, kinds)
    torch.cuda.synchronize()
    res = device.results_to_host(d_res, n)
    got = d_out.cpu().numpy()
    mask = np.ones(out_total, dtype=bool)
    for o, cap in zip(out_offs, caps):
        mask[o:o + cap] = False
    bad = np.nonzero(mask & (got != OUT_POISON))[0]
    detail = ""
    if len(bad):
        first = int(bad[0])
        k = int(np.searchsorted(np.array(out_offs), first, side="right")) - 1
        detail = f"{len(bad)} poisoned bytes overwritten; first at arena offset {first}, near job {k}'s output slot (cap {caps[max(k, 0)]})"
    if not np.array_equal(d_in.cpu().numpy(), h_in):
        detail += " | the INPUT arena was written to"
    out = []
    for i in range(n):
        st = int(res["status"][i])
        out.append((st, got[out_offs[i]:out_offs[i] + int(res["out_len"][i])].tobytes() if st == ffi.OK else b""))
    return out, detail == "", detail


def check_decompress(items, expect=None, label="", max_input_len=None):
    """Both IN-poisons: zones intact, statuses and Ok bytes identical under both, and (when given) equal to `expect`
    [(status, bytes) from the oracle].  Returns the results."""
    ra, ok_a, da = decompress_guarded(items, 0x00, seed=1, max_input_len=max_input_len)
    rb, ok_b, db = decompress_guarded(items, 0xFF, seed=2, max_input_len=max_input_len)
    assert ok_a, f"{label}: red zone violated (input poison 0x00): {da}"
    assert ok_b, f"{label}: red zone violated (input poison 0xFF): {db}"
    for i, ((sa, ba), (sb, bb)) in enumerate(zip(ra, rb)):
        assert sa == sb, f"{label}: job {i} status depends on the bytes behind input_len ({sa} vs {sb})"
        if sa == 0:
            assert ba == bb, f"{label}: job {i} output depends on the bytes behind input_len"
    if expect is not None:
        for i, ((s, b), (es, eb)) in enumerate(zip(ra, expect)):
            assert s == es, f"{label}: job {i} status {s}, oracle {es}"
            if s == 0:
                assert b == eb, f"{label}: job {i} bytes differ from the oracle's"
    return ra


def check_compress(items, expect=None, label=""):
    ra, ok_a, da = compress_guarded(items, 0x00, seed=3)
    rb, ok_b, db = compress_guarded(items, 0xFF, seed=4)
    assert ok_a, f"{label}: red zone violated (input poison 0x00): {da}"
    assert ok_b, f"{label}: red zone violated (input poison 0xFF): {db}"
    for i, (a, b) in enumerate(zip(ra, rb)):
        assert a == b, f"{label}: job {i} result depends on the bytes behind input_len"
    if expect is not None:
        for i, ((s, b), (es, eb)) in enumerate(zip(ra, expect)):
            assert s == es and (s != 0 or b == eb), f"{label}: job {i} differs from the oracle ({s} vs {es})"
    return ra
