#!/usr/bin/env python3
"""bench.py — GiB/s of the LZ4 raw-block hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload, BASELINE.json configs[1]): the Silesia stand-in `silesia_mix`
(211 938 580 B, rust-lz-fear_amd/synth.py) cut into 4 MiB independent blocks, tiled `--copies`
times per GPU so that one launch has enough independent blocks to occupy 256 CUs; a "step" is
one decompress pass over every block this rank owns (kernel for compressed blocks + the
device copy the frame layer does for stored blocks).  Inputs are resident in HBM before the
timed region.  Blocks shard across ranks with no data-path collective ("weak" scaling: per-GPU
work fixed).  The compress pass over the same blocks (configs[2]) is timed too and reported
under "compress".
"""
import argparse
import concurrent.futures
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import device, ffi, synth  # noqa: E402

BS = 4 << 20
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def timed_launches(fn, steps):
    """Run fn() `steps` times; HIP events on the launch stream bracket each call."""
    evs = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    return evs


def decompress_kernel_name(n_jobs):
    """What lzf_decompress_batch launches for a batch of n_jobs (capi.hip decompress_variant; 256 CUs)."""
    v = os.environ.get("LZF_DECOMPRESS_KERNEL", "auto")
    if v != "auto":
        return v
    if n_jobs <= 2048:
        return "lzf_decompress_paired_kernel<4096,48,640>"
    return "lzf_decompress_paired_kernel<4096,24,384>" if n_jobs <= 16384 else "lzf_decompress_batched_kernel<4096,16,256,staged>"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--copies", type=int, default=240, help="tiled copies of silesia_mix per GPU (51 blocks each)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:     # under torch.distributed.run: always the RCCL path
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", device_id=dev)
        dist = dist_mod
    assert ffi.device_count() >= 1

    # ------------------------------------------------------------------ inputs (synthetic)
    t0 = time.time()
    base = torch.from_numpy(synth.silesia_mix()).to(dev)
    total = base.numel()
    nb1 = (total + BS - 1) // BS                      # 51 blocks per copy, the last one short
    copies = args.copies
    nblk = nb1 * copies
    # every copy gets its own slab of nb1 * BS bytes (block-aligned); copy k > 0 = base ^ c_k so
    # that no two copies hold the same bytes (global copy index differs per rank)
    src = torch.zeros(nblk * BS, dtype=torch.uint8, device=dev)
    lens = np.full(nblk, BS, dtype=np.uint64)
    for k in range(copies):
        g = rank * copies + k
        c = (g * 37 + (g >> 3)) & 0xFF if g else 0
        dst = src[k * nb1 * BS:k * nb1 * BS + total]
        torch.bitwise_xor(base, c, out=dst) if c else dst.copy_(base)
        lens[k * nb1 + nb1 - 1] = total - (nb1 - 1) * BS
    del base
    torch.cuda.synchronize()
    log(f"[bench] rank {rank}: {nblk} blocks, {src.numel() / 2**30:.2f} GiB source in HBM ({time.time() - t0:.1f}s)")

    # ------------------------------------------------------------------ compress (configs[2])
    comp = torch.empty(nblk * BS, dtype=torch.uint8, device=dev)      # slot stride = block size (cap = N)
    cj = np.zeros(nblk, dtype=device.CJOB)
    cj["input"] = np.uint64(src.data_ptr()) + np.arange(nblk, dtype=np.uint64) * np.uint64(BS)
    cj["input_len"] = lens
    cj["out"] = np.uint64(comp.data_ptr()) + np.arange(nblk, dtype=np.uint64) * np.uint64(BS)
    cj["out_cap"] = lens                                               # framed/compress.rs:242
    cj["table_kind"] = ffi.TABLE_U32
    d_cj = device.to_device(cj, dev)
    d_cres = torch.zeros(nblk * 16, dtype=torch.uint8, device=dev)

    def compress_step():
        device.compress_batch(d_cj, d_cres, nblk, ffi.KINDS_U32)

    c_steps = max(1, min(args.steps, 3))
    compress_step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    tc0 = time.perf_counter()
    c_evs = timed_launches(compress_step, c_steps)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    tc = time.perf_counter() - tc0
    cres = device.results_to_host(d_cres, nblk)
    ok = cres["status"] == ffi.OK
    full = cres["status"] == ffi.OUTPUT_FULL
    assert np.all(ok | full), f"compress statuses: {np.unique(cres['status'])}"
    clen = np.where(ok, cres["out_len"], 0).astype(np.uint64)
    c_kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in c_evs]))
    c_bytes = float(lens.sum() + clen.sum())                           # N + C

    # stored blocks (incompressible, framed/compress.rs:250-255): the frame carries the raw bytes
    stored_idx = np.nonzero(full)[0]
    comp2d, src2d = comp.view(nblk, BS), src.view(nblk, BS)
    if len(stored_idx):
        t_idx = torch.from_numpy(stored_idx.astype(np.int64)).to(dev)
        comp2d[t_idx] = src2d[t_idx]
        # device-side move of the stored blocks during decompression (lzf_copy_ranges, on a second stream so that it
        # overlaps the decompress kernel)
        sidx = stored_idx.astype(np.uint64)
        d_sp = torch.from_numpy((np.uint64(comp.data_ptr()) + sidx * np.uint64(BS)).view(np.int64)).to(dev)
        d_dp_base = sidx * np.uint64(BS)
        d_sl = torch.from_numpy(lens[stored_idx].astype(np.uint64).view(np.int64)).to(dev)
        side = torch.cuda.Stream(device=dev)
    else:
        t_idx = None

    # ------------------------------------------------------------------ decompress (configs[1])
    dec = torch.zeros(nblk * BS, dtype=torch.uint8, device=dev)
    dec2d = dec.view(nblk, BS)
    kidx = np.nonzero(ok)[0]
    if os.environ.get("LZF_BENCH_SORT", "0") == "1":
        # analysis knob: longest-compressed-first job array built here.  Not needed any more: lzf_decompress_batch and
        # lzf_compress_batch order large batches on the device themselves (LZF_DECOMPRESS_ORDER / LZF_COMPRESS_ORDER).
        kidx = kidx[np.argsort(-clen[kidx].astype(np.int64), kind="stable")]
    nk = len(kidx)
    dj = np.zeros(nk, dtype=device.DJOB)
    dj["input"] = np.uint64(comp.data_ptr()) + kidx.astype(np.uint64) * np.uint64(BS)
    dj["input_len"] = clen[kidx]
    dj["out"] = np.uint64(dec.data_ptr()) + kidx.astype(np.uint64) * np.uint64(BS)
    dj["out_cap"] = lens[kidx]
    dj["output_limit"] = BS                                            # block_maxsize, framed/decompress.rs:248
    d_dj = device.to_device(dj, dev)
    d_dres = torch.zeros(max(nk, 1) * 16, dtype=torch.uint8, device=dev)

    def decompress_kernel():
        device.decompress_batch(d_dj, d_dres, nk)

    if t_idx is not None:
        d_dp = torch.from_numpy((np.uint64(dec.data_ptr()) + d_dp_base).view(np.int64)).to(dev)

    def decompress_step():
        if t_idx is not None:                                          # framed/decompress.rs:250, next to the kernel
            side.wait_stream(torch.cuda.current_stream())
            device.copy_ranges(d_sp, d_dp, d_sl, len(stored_idx), BS, stream=side)
        evs.extend(timed_launches(decompress_kernel, 1))
        if t_idx is not None:
            torch.cuda.current_stream().wait_stream(side)

    evs = []
    for _ in range(args.warmup):
        decompress_step()
    evs.clear()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        decompress_step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    d_kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    dres = device.results_to_host(d_dres, nk)
    assert np.all(dres["status"] == ffi.OK), f"decompress statuses: {np.unique(dres['status'])}"
    assert np.array_equal(dres["out_len"], lens[kidx])
    if not args.no_verify:
        assert torch.equal(dec, src), "decoded bytes differ from the source"   # round trip at full size

    t = torch.tensor([elapsed, tc], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(lens.sum())], dtype=torch.float64, device=dev)
    if dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed_max, tc_max = float(t[0]), float(t[1])
    total_bytes = float(tot[0])

    d_bytes = float(lens[kidx].sum() + clen[kidx].sum())               # C + N of the kernel's jobs
    # HBM traffic from the PMC passes committed under profiles/ (FETCH_SIZE doubled per the gfx950
    # correction + WRITE_SIZE), scaled from the profiled launch to this launch by job count
    d_traffic = c_traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        d_traffic = (2 * tj["decompress"]["FETCH_SIZE_KB"] + tj["decompress"]["WRITE_SIZE_KB"]) * 1024.0 * nk / tj["decompress"]["jobs"]
        c_traffic = (2 * tj["compress"]["FETCH_SIZE_KB"] + tj["compress"]["WRITE_SIZE_KB"]) * 1024.0 * nblk / tj["compress"]["jobs"]
    d_achieved = d_bytes / (d_kernel_ms * 1e-3) / 1e9
    c_achieved = c_bytes / (c_kernel_ms * 1e-3) / 1e9

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(comp2d, clen, lens, kidx, args.cpu_seconds)

    if rank == 0:
        value = total_bytes * args.steps / elapsed_max / 2**30
        line = {
            "metric": "GiB/s compress + decompress, 4 MiB independent blocks, 1/2/4/8 MI355X",
            "value": round(value, 3), "unit": "GiB/s (uncompressed bytes decompressed per second)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "silesia_mix (Silesia stand-in, 211938580 B) x %d copies per GPU, 4 MiB independent "
                                   "blocks, decompress_raw of every block (configs[1]); lz4 ratio %.3f" %
                                   (copies, float(lens.sum()) / float(clen.sum() + lens[stored_idx].sum())),
                       "blocks_per_gpu": int(nblk), "stored_blocks_per_gpu": int(len(stored_idx)),
                       "block_size": BS, "parallelism": f"block-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "kernel": decompress_kernel_name(nk),
                         "achieved": round(d_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(d_achieved / HBM_PEAK_GBS, 5), "traffic": d_traffic,
                         "algorithmic_bytes_per_launch": d_bytes, "kernel_ms": round(d_kernel_ms, 4)},
            "compress": {"value": round(total_bytes * c_steps / tc_max / 2**30, 3), "unit": "GiB/s (uncompressed bytes compressed per second)",
                         "steps": c_steps, "ms_per_step": round(tc_max / c_steps * 1e3, 3),
                         "roofline": {"bound": "hbm", "kernel": "lzf_compress_compact_kernel",
                                      "achieved": round(c_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(c_achieved / HBM_PEAK_GBS, 5), "traffic": c_traffic,
                                      "algorithmic_bytes_per_launch": c_bytes, "kernel_ms": round(c_kernel_ms, 3)}},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


def cpu_baseline(comp2d, clen, lens, kidx, budget_s):
    """The CPU restatement of lz-fear (oracle/, kind "port": the Rust reference cannot be built
    here) timed on this host's cores over a bounded sample of the same blocks."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oracle_ffi as o
    L = o.lib()
    cores = os.cpu_count() or 1
    threads = min(cores, 32)
    sample = kidx[: max(threads, min(len(kidx), 48))]
    host = [(comp2d[int(i), : int(clen[i])].cpu().numpy().tobytes(), int(lens[i])) for i in sample]
    # ---- decompress
    def dec_one(item):
        c, n = item
        out = C.create_string_buffer(n + 64)
        ln = C.c_size_t(0)
        rc = L.lzfo_decompress_raw(c, len(c), b"", 0, out, C.byref(ln), n + 64, n)
        assert rc == 0 and ln.value == n
        return out
    srcs = []
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(threads) as ex:
        outs = list(ex.map(dec_one, host))
    t_once = time.perf_counter() - t0
    srcs = [o_.raw[:n] for o_, (_, n) in zip(outs, host)]
    reps = max(1, int(budget_s * 0.4 / max(t_once, 1e-3)))
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(threads) as ex:
        for _ in range(reps):
            list(ex.map(dec_one, host))
    t_dec = (time.perf_counter() - t0) / reps
    nbytes = sum(n for _, n in host)
    # ---- compress (same blocks)
    def comp_one(s):
        t = o.U32Table()
        out = C.create_string_buffer(len(s) + 64)
        ln = C.c_size_t(0)
        L.lzfo_compress2(s, len(s), 0, 0, C.addressof(t), out, len(s), C.byref(ln))
        return ln.value
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(threads) as ex:
        list(ex.map(comp_one, srcs))
    t_comp = time.perf_counter() - t0
    return {"value": round(nbytes / t_dec / 2**30, 3), "unit": "GiB/s (decompress, uncompressed bytes)",
            "cores": threads, "kind": "port",
            "sample": f"{len(host)} of the same 4 MiB blocks ({nbytes / 2**20:.0f} MiB), {threads} threads, "
                      f"{reps} reps; oracle/lzf_oracle.c (lz-fear restated in C, gcc -O3)",
            "compress_value": round(nbytes / t_comp / 2**30, 3), "host_cpus": cores}


if __name__ == "__main__":
    main()
