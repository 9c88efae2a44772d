"""lzf_frame_gather (include/lzfear_dist.h) with MORE THAN ONE RANK: the ncclAllGather of the size table and the grouped ncclSend /
ncclRecv of the segments, which a one-GPU box never executes.  Run by tests/test_gpu_frame.py::test_frame_gather_c_abi_rccl_world2
as WORLD_SIZE processes (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment, one GPU per rank) when the box has at
least two GPUs.  Every case is checked on EVERY rank against the oracle's frame; the failure cases must fail on every rank alike."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import oracle_ffi as o  # noqa: E402
import vectors  # noqa: E402
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import device, ffi, synth, dist as lzdist  # noqa: E402

BS = 64 << 10


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", device_id=dev)
    comm = lzdist.DistComm(dist, rank, world, dev)
    assert comm.count() == world, (comm.count(), world)
    header = lzdist.frame_header(content_checksum=False, block_size=BS)
    paths = lzdist.rccl_paths()

    def stream(n_blocks, last_len, stored_at):
        parts = []
        for b in range(n_blocks):
            ln = last_len if b == n_blocks - 1 else BS
            parts.append(np.frombuffer(vectors.rng_bytes(100 + b, ln), np.uint8) if b == stored_at else synth.log_text(b * BS, b * BS + ln))
        return np.concatenate(parts)

    def run(n_blocks, last_len, stored_at, spoil=None):
        data = stream(n_blocks, last_len, stored_at)
        lo, hi = lzdist.shard_range(n_blocks, rank, world)
        nloc = hi - lo
        mine = data[lo * BS:min(hi * BS, len(data))]
        src = torch.zeros(max(nloc, 1) * BS, dtype=torch.uint8, device=dev)
        src[:len(mine)] = torch.from_numpy(mine.copy()).to(dev)
        comp = torch.empty(max(nloc, 1) * BS, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(max(nloc, 1) * 16, dtype=torch.uint8, device=dev)
        if nloc:
            cj = np.zeros(nloc, dtype=device.CJOB)
            cj["input"] = np.uint64(src.data_ptr()) + np.arange(nloc, dtype=np.uint64) * np.uint64(BS)
            cj["input_len"] = [last_len if lo + i == n_blocks - 1 else BS for i in range(nloc)]
            cj["out"] = np.uint64(comp.data_ptr()) + np.arange(nloc, dtype=np.uint64) * np.uint64(BS)
            cj["out_cap"] = cj["input_len"]
            cj["table_kind"] = ffi.TABLE_U32
            device.compress_batch(device.to_device(cj, dev), d_res, nloc, ffi.KINDS_U32 | ffi.KINDS_U32_FRESH_ONLY)
            torch.cuda.synchronize()
        frame = torch.zeros(64 + n_blocks * (BS + 8), dtype=torch.uint8, device=dev)
        hdr, fr = header, frame
        if spoil == "status" and rank == world - 1 and nloc:
            r = device.results_to_host(d_res, nloc).copy(); r["status"][0] = ffi.CONTRACT
            d_res = device.to_device(r, dev)
        if spoil == "cap" and rank == 0:
            fr = frame[:100]
        if spoil == "header" and rank == world - 1:
            hdr = lzdist.frame_header(content_checksum=True, block_size=BS)
        try:
            flen, ctot = lzdist.gather_frame_device_c(comm, d_res, comp, src, BS, nloc, n_blocks, fr, hdr, last_block_len=last_len)
        except ffi.LzfError as e:
            return ("error", e.code)
        got = frame[:flen].cpu().numpy().tobytes()
        rc, ref = o.frame_compress(data.tobytes(), o.make_settings(block_size=BS, content_checksum=False))
        assert rc == 0 and got == ref, (rank, n_blocks, "the frame differs from the oracle's")
        return ("ok", flen)

    results = []
    for n_blocks, last_len, stored_at in ((7, BS, 3), (8, 12_345, 0), (1, 777, -1), (2 * world + 1, BS, 2 * world)):
        results.append(run(n_blocks, last_len, stored_at))
        assert results[-1][0] == "ok", (rank, n_blocks, results[-1])
    # failures local to ONE rank: every rank must return (no hang) with the same code
    for spoil, want in (("status", ffi.CONTRACT), ("cap", ffi.E_INVALID), ("header", ffi.E_INVALID)):
        r = run(8, BS, 5, spoil=spoil)
        assert r == ("error", want), (rank, spoil, r)
        codes = [None] * world
        dist.all_gather_object(codes, r)
        assert all(c == r for c in codes), (spoil, codes)
    # ... and the communicator still works afterwards
    assert run(8, BS, 1)[0] == "ok"
    comm.close()
    dist.barrier()
    if rank == 0:
        print("world2 ok:", world, "ranks;", paths)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
