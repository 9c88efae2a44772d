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
