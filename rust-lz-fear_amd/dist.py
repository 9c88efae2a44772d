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
    ptrs = (C.c_void_p * n)(*[C.cast(b, C.c_void_p) for b in bufs])
    cl = (C.c_uint32 * n)(*[int(c) for c in comp_len])
    rl = (C.c_uint32 * n)(*[int(r) for r in raw_len])
    cap = 64 + sum(len(h) + 8 for h in host)
    out = C.create_string_buffer(cap)
    outlen = C.c_size_t(0)
    rc = ffi.lib().lzf_frame_assemble(C.byref(settings), n, ptrs, cl, rl, content_xxh32, out, cap, C.byref(outlen))
    if rc != 0:
        raise ffi.LzfError(rc, "lzf_frame_assemble failed")
    return out.raw[: outlen.value]
