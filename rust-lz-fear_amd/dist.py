"""Block-range sharding of one frame across ranks and reassembly (BASELINE config 4).

In independent-blocks mode every block is a self-contained job (src/framed/compress.rs:265-270), so
rank r of W compresses the contiguous block range `shard_range(n_blocks, r, W)` with no
data-path communication.  Reassembling ONE frame needs exactly one exchange: the per-block u32
size words, then the variable-size payloads, ordered by block index.  Both are all-gathers over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests).
The codec itself is never involved here — this is plumbing around lzf_frame_assemble."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import ffi

STORED = 0xFFFFFFFF      # size-table marker: block kept raw (compress2 -> OutputFull)


def shard_range(n_blocks, rank, world):
    """Contiguous block range [lo, hi) of `rank`; the first n_blocks % world ranks get one more."""
    base, extra = divmod(n_blocks, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allgather_blocks(local_payloads, local_comp_len, n_blocks, device="cpu", group=None):
    """All-gather the blocks every rank produced.

    local_payloads : list of uint8 tensors (compressed bytes, or the raw block when stored)
    local_comp_len : list of ints, STORED for raw blocks
    Returns (payload list of uint8 tensors in block order, comp_len np.uint32 array) on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n_blocks, rank, world)
    assert len(local_payloads) == hi - lo == len(local_comp_len)
    max_blocks = (n_blocks + world - 1) // world
    # 1. size table: [comp_len, payload_bytes] per block, padded to max_blocks rows
    tab = torch.zeros((max_blocks, 2), dtype=torch.int64, device=device)
    for i, (p, c) in enumerate(zip(local_payloads, local_comp_len)):
        tab[i, 0] = int(c)
        tab[i, 1] = p.numel()
    tabs = [torch.empty_like(tab) for _ in range(world)]
    dist.all_gather(tabs, tab, group=group)
    # 2. payload slab, padded to the largest per-rank byte count
    per_rank_bytes = [int(t[:, 1].sum()) for t in tabs]
    slab = torch.zeros(max(max(per_rank_bytes), 1), dtype=torch.uint8, device=device)
    off = 0
    for p in local_payloads:
        slab[off:off + p.numel()] = p.to(device)
        off += p.numel()
    slabs = [torch.empty_like(slab) for _ in range(world)]
    dist.all_gather(slabs, slab, group=group)
    payloads, comp_len = [], []
    for r in range(world):
        rlo, rhi = shard_range(n_blocks, r, world)
        off = 0
        for i in range(rhi - rlo):
            nbytes = int(tabs[r][i, 1])
            payloads.append(slabs[r][off:off + nbytes])
            comp_len.append(int(tabs[r][i, 0]))
            off += nbytes
    return payloads, np.array(comp_len, dtype=np.uint32)


def assemble_frame(settings, payloads, comp_len, raw_len, content_xxh32):
    """lzf_frame_assemble over gathered blocks -> frame bytes (host)."""
    n = len(payloads)
    host = [bytes(p.cpu().numpy().tobytes()) for p in payloads]
    bufs = [C.create_string_buffer(h, max(len(h), 1)) for h in host]
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    cl = (C.c_uint32 * n)(*[int(c) for c in comp_len])
    rl = (C.c_uint32 * n)(*[int(r) for r in raw_len])
    cap = 64 + sum(len(h) + 8 for h in host)
    out = C.create_string_buffer(cap)
    outlen = C.c_size_t(0)
    rc = ffi.lib().lzf_frame_assemble(C.byref(settings), n, ptrs, cl, rl, content_xxh32, out, cap, C.byref(outlen))
    if rc != 0:
        raise ffi.LzfError(rc, "lzf_frame_assemble failed")
    return out.raw[: outlen.value]


# ------------------------------------------------------------------------------------------------------------------
# Device path (BASELINE config 4, bench.py --workload config4): everything stays in HBM, no per-block Python.
# A rank's part of the frame body — [u32 size word][payload] per block of its range — is contiguous in the frame,
# so every rank packs its own segment straight into the frame buffer at its final offset and the exchange is one
# exact-size send / receive per peer pair (the direct all-gather xGMI offers: every link carries one segment).
# ------------------------------------------------------------------------------------------------------------------
def frame_header(content_checksum=False, block_size=4 << 20, independent=True, block_checksums=False):
    """Header bytes of a frame without content size / dictionary id (src/framed/compress.rs:163-200, header.rs:30-81):
    magic, FLG, BD, HC = second byte of XXH32(FLG BD)."""
    code = {64 << 10: 4, 256 << 10: 5, 1 << 20: 6, 4 << 20: 7}[block_size]
    flg = (1 << 6) | (int(independent) << 5) | (int(block_checksums) << 4) | (int(content_checksum) << 2)
    desc = bytes([flg, code << 4])
    hc = (ffi.lib().lzf_xxh32(desc, len(desc), 0) >> 8) & 0xFF
    return bytes([0x04, 0x22, 0x4D, 0x18]) + desc + bytes([hc])


def gather_frame_device(d_cres, comp, src, block_size, n_local, n_blocks, frame, dist_mod, rank, world, device_mod, header, raw_lens=None):
    """Pack this rank's compressed (or stored) blocks into `frame` (uint8 CUDA tensor) at their final offsets, exchange the
    segments, write header and EndMark.  d_cres: the lzf_job_result array of the rank's compress launch (HBM); comp: the
    compressed slots (stride block_size); src: the rank's raw blocks; raw_lens: int64 CUDA tensor of the blocks' raw lengths
    (None: every block is block_size long — only the stream's last block may be shorter).
    Returns (frame bytes, total compressed payload bytes)."""
    dev = frame.device
    res = d_cres.view(torch.int64).view(-1, 2)[:n_local]
    out_len = res[:, 0]
    status = res[:, 1] & 0xFFFFFFFF
    okm = status == ffi.OK                                        # else OutputFull: stored raw (framed/compress.rs:250-255)
    if not bool(torch.all(okm | (status == ffi.OUTPUT_FULL))):   # anything else is not "store it raw" (compress.rs:250-255 asserts the kind)
        raise RuntimeError(f"compress statuses {torch.unique(status).tolist()}: only OK / OUTPUT_FULL can be framed")
    rawl = torch.full_like(out_len, block_size) if raw_lens is None else raw_lens.to(torch.int64)
    plen = torch.where(okm, out_len, rawl)
    word = torch.where(okm, out_len, rawl | 0x80000000)
    max_blocks = (n_blocks + world - 1) // world
    tab = torch.zeros(max_blocks, dtype=torch.int64, device=dev)
    tab[:n_local] = plen
    if world > 1:
        tabs = torch.empty(world * max_blocks, dtype=torch.int64, device=dev)
        dist_mod.all_gather_into_tensor(tabs, tab)                 # the size table
        tabs = tabs.view(world, max_blocks)
    else:
        tabs = tab.view(1, max_blocks)
    seg = (tabs + 4 * (tabs > 0)).sum(dim=1)                       # bytes of every rank's segment (a block's payload is never empty)
    seg_host = seg.cpu().tolist()                                  # (one small device -> host copy: tensor slices need Python ints)
    hl = len(header)
    seg_off = [hl]
    for r in range(world):
        seg_off.append(seg_off[-1] + int(seg_host[r]))
    flen = seg_off[-1] + 4
    assert flen <= frame.numel(), "frame buffer too small"
    # ---- this rank's segment, in place
    base = seg_off[rank]
    boff = base + torch.cumsum(plen + 4, 0) - (plen + 4)           # where block i's size word goes
    idx = (boff[:, None] + torch.arange(4, device=dev)[None, :]).reshape(-1)
    wb = ((word[:, None] >> (8 * torch.arange(4, device=dev))[None, :]) & 0xFF).to(torch.uint8).reshape(-1)
    frame[idx] = wb
    i64 = torch.arange(n_local, dtype=torch.int64, device=dev)
    sp = torch.where(okm, comp.data_ptr() + i64 * block_size, src.data_ptr() + i64 * block_size)
    dp = frame.data_ptr() + boff + 4
    device_mod.copy_ranges(sp, dp, plen, n_local, block_size)
    # ---- exchange: my segment to every peer, theirs into their places
    if world > 1:
        ops = []
        mine = frame[seg_off[rank]:seg_off[rank + 1]]
        for p in range(world):
            if p == rank:
                continue
            ops.append(dist_mod.P2POp(dist_mod.isend, mine, p))
            ops.append(dist_mod.P2POp(dist_mod.irecv, frame[seg_off[p]:seg_off[p + 1]], p))
        for w in dist_mod.batch_isend_irecv(ops):
            w.wait()
    frame[:hl] = torch.tensor(list(header), dtype=torch.uint8, device=dev)
    frame[flen - 4:flen] = 0                                       # EndMark (no content checksum in this mode)
    comp_total = int(sum(seg_host)) - 4 * n_blocks
    return flen, comp_total


# ------------------------------------------------------------------------------------------------------------------
# The same exchange under the C ABI (include/lzfear_dist.h, liblzfear_dist.so: ncclAllGather of the size words, in-place packing,
# grouped ncclSend / ncclRecv of the exact-size segments) — what a Rust / C host calls.  Python only bootstraps the communicator:
# rank 0's ncclUniqueId travels over the process group that already exists.
# ------------------------------------------------------------------------------------------------------------------
_dist_lib = None


def dist_lib():
    """liblzfear_dist.so (built by __graft_entry__.build(); links librccl — in a torch process the loader binds it to the RCCL torch loaded)."""
    global _dist_lib
    if _dist_lib is None:
        import os
        from . import build as _build
        ffi.lib()                                             # liblzfear_hip.so first: the dist library needs it
        path = _build.DIST_LIB_PATH
        if not os.path.exists(path):
            raise ffi.LzfError(ffi.E_INVALID, f"{path} is missing — run __graft_entry__.build()")
        L = C.CDLL(path)
        L.lzf_dist_last_error.restype = C.c_char_p
        L.lzf_dist_unique_id.argtypes = [C.c_char_p]
        L.lzf_dist_comm_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.lzf_dist_comm_count.argtypes = [C.c_void_p]
        L.lzf_dist_comm_free.argtypes = [C.c_void_p]
        L.lzf_dist_rccl_path.restype = C.c_char_p
        L.lzf_frame_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64,
                                       C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]
        _dist_lib = L
    return _dist_lib


def rccl_paths():
    """Which librccl the exchange library's calls resolve to (dladdr inside liblzfear_dist.so) and every librccl mapped into this
    process (/proc/self/maps: torch carries its own copy) — bench.py prints both, so a run shows whether they are the same file."""
    mine = dist_lib().lzf_dist_rccl_path().decode()
    mapped = set()
    try:
        for line in open("/proc/self/maps"):
            f = line.rstrip("\n").split(None, 5)
            if len(f) == 6 and "librccl" in f[5]:
                mapped.add(f[5])
    except OSError:
        pass
    return {"liblzfear_dist_binds": mine, "mapped_in_process": sorted(mapped), "single_copy": len(mapped) <= 1}


class DistComm:
    """lzf_dist_comm: an RCCL communicator made through the C ABI.  `dist_mod` (torch.distributed, initialised) only carries rank 0's
    ncclUniqueId to the other ranks; None = a single rank."""

    def __init__(self, dist_mod, rank, world, device):
        L = dist_lib()
        uid = C.create_string_buffer(128)
        rc0, err0 = 0, ""
        if rank == 0:
            rc0 = L.lzf_dist_unique_id(uid)
            if rc0 != 0:
                err0 = L.lzf_dist_last_error().decode()
        if world > 1:
            # rank 0's outcome travels WITH the id (byte 128): a failure there must not leave the other ranks in the broadcast
            # (ADVICE r5), every rank raises instead
            t = torch.frombuffer(bytearray(uid.raw) + bytearray([1 if rc0 == 0 else 0]), dtype=torch.uint8).to(device)
            dist_mod.broadcast(t, src=0)
            got = bytes(t.cpu().numpy().tobytes())
            uid = C.create_string_buffer(got[:128], 128)
            if got[128] != 1:
                raise ffi.LzfError(rc0 or ffi.E_HIP, "lzf_dist_unique_id failed on rank 0" + (": " + err0 if err0 else ""))
        elif rc0 != 0:
            raise ffi.LzfError(rc0, err0)
        self.handle = C.c_void_p()
        rc = L.lzf_dist_comm_init(uid, rank, world, C.byref(self.handle))
        if rc != 0:
            raise ffi.LzfError(rc, L.lzf_dist_last_error().decode())
        self.rank, self.world = rank, world

    def count(self):
        return dist_lib().lzf_dist_comm_count(self.handle)

    def close(self):
        if self.handle:
            dist_lib().lzf_dist_comm_free(self.handle)
            self.handle = C.c_void_p()


def gather_frame_device_c(comm, d_cres, comp, src, block_size, n_local, n_blocks, frame, header, last_block_len=None, stream=None):
    """gather_frame_device through lzf_frame_gather (include/lzfear_dist.h).  Returns (frame bytes, total compressed payload bytes)."""
    flen, ctot = C.c_uint64(0), C.c_uint64(0)
    st = (stream or torch.cuda.current_stream()).cuda_stream
    rc = dist_lib().lzf_frame_gather(comm.handle, d_cres.data_ptr(), comp.data_ptr(), src.data_ptr(), block_size, block_size, n_local, n_blocks,
                                     block_size if last_block_len is None else last_block_len, bytes(header), len(header), frame.data_ptr(), frame.numel(),
                                     C.byref(flen), C.byref(ctot), st)
    if rc != 0:
        raise ffi.LzfError(rc, dist_lib().lzf_dist_last_error().decode())
    return int(flen.value), int(ctot.value)
