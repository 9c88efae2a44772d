"""CPU test of the N > 1 path (world_size 2, gloo): block-range sharding of one frame, the
size-table + payload all-gather, and frame assembly — must reproduce the single-process frame
byte for byte.  The per-rank block payloads come from the oracle here (no GPU in this container);
on the GPU node the same plumbing carries lzf_compress_batch output."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_ffi as o
import rust_lz_fear_amd  # noqa: F401
from rust_lz_fear_amd import build, dist as lzdist, ffi, framed, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bs, data, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_blocks = (len(data) + bs - 1) // bs
        lo, hi = lzdist.shard_range(n_blocks, rank, world)
        payloads, clens = [], []
        for b in range(lo, hi):
            blk = data[b * bs:(b + 1) * bs]
            rc, comp = o.compress2(blk, cap=len(blk))          # stand-in for the GPU batch on this rank
            if rc == 0:
                payloads.append(torch.frombuffer(bytearray(comp), dtype=torch.uint8)); clens.append(len(comp))
            else:
                payloads.append(torch.frombuffer(bytearray(blk), dtype=torch.uint8)); clens.append(lzdist.STORED)
        allp, allc = lzdist.allgather_blocks(payloads, clens, n_blocks)
        raw_len = [min(bs, len(data) - b * bs) for b in range(n_blocks)]
        s = framed.CompressionSettings().block_size(bs)._struct(None)
        frame = lzdist.assemble_frame(s, allp, allc, raw_len, ffi.lib().lzf_xxh32(data, len(data), 0))
        q.put((rank, frame))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for n in (0, 1, 2, 7, 51, 256, 2048):
        for w in (1, 2, 3, 4, 8):
            spans = [lzdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


@pytest.mark.timeout(300)
def test_two_rank_frame_reassembly_matches_single_process_frame():
    build.build_library()
    bs = 64 << 10
    data = synth.silesia_mix(29 << 20, (29 << 20) + 7 * bs + 12345).tobytes() + synth.gen_random(3, 2 * bs).tobytes()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bs, data, q)) for r in range(2)]
    for p in procs:
        p.start()
    frames = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rc, want = o.frame_compress(data, o.make_settings(block_size=bs))
    assert rc == 0
    assert frames[0] == want and frames[1] == want
    assert o.frame_decompress(frames[0])[:2] == (0, data)


# ---- the device-side gather of bench.py --workload config4 (dist.gather_frame_device), world_size 2 on gloo: the size-table
#      all-gather, the in-place packing at final offsets and the exact-size segment exchange are plain torch.distributed calls;
#      only the byte mover (lzf_copy_ranges on the GPU) is replaced by a memmove loop here.
class _CpuMover:
    @staticmethod
    def copy_ranges(sp, dp, plen, n, max_len, stream=None):
        import ctypes
        for s, d, k in zip(sp.tolist(), dp.tolist(), plen.tolist()):
            assert 0 < k <= max_len
            ctypes.memmove(d, s, k)


def _gather_worker(rank, world, port, bs, data, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_blocks = len(data) // bs
        lo, hi = lzdist.shard_range(n_blocks, rank, world)
        n_local = hi - lo
        src = torch.frombuffer(bytearray(data[lo * bs: hi * bs]), dtype=torch.uint8)
        comp = torch.zeros(n_local * bs, dtype=torch.uint8)
        res = np.zeros(n_local, dtype=[("out_len", "<u8"), ("status", "<i4"), ("reserved", "<u4")])
        for i in range(n_local):
            blk = data[(lo + i) * bs:(lo + i + 1) * bs]
            rc, c = o.compress2(blk, cap=len(blk))                # stand-in for lzf_compress_batch on this rank
            res["status"][i] = rc
            if rc == 0:
                res["out_len"][i] = len(c)
                comp[i * bs: i * bs + len(c)] = torch.frombuffer(bytearray(c), dtype=torch.uint8)
        d_cres = torch.from_numpy(res.view(np.uint8).copy())
        frame = torch.zeros(64 + n_blocks * (bs + 8), dtype=torch.uint8)
        header = lzdist.frame_header(content_checksum=False, block_size=bs)
        flen, comp_total = lzdist.gather_frame_device(d_cres, comp, src, bs, n_local, n_blocks, frame, dist, rank, world, _CpuMover, header)
        q.put((rank, frame[:flen].numpy().tobytes(), comp_total))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n_blocks", [7, 8])
def test_two_rank_in_place_frame_gather_matches_single_process_frame(n_blocks):
    build.build_library()
    bs = 64 << 10
    data = (synth.silesia_mix(31 << 20, (31 << 20) + (n_blocks - 2) * bs).tobytes() + synth.gen_random(5, bs).tobytes() +
            synth.log_text(0, bs).tobytes())                     # one stored block in rank 1's range
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, bs, data, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rc, want = o.frame_compress(data, o.make_settings(block_size=bs, content_checksum=False))
    assert rc == 0
    for rank, frame, comp_total in got:
        assert frame == want, rank
        assert comp_total == len(want) - 7 - 4 - 4 * n_blocks
    assert o.frame_decompress(want)[:2] == (0, data)
