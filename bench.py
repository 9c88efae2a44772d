#!/usr/bin/env python3
"""bench.py — GiB/s of the LZ4 raw-block hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--workload silesia|config4]
  N > 1 without a launcher (no WORLD_SIZE in the environment): the script starts itself under
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` — one rank per GPU
  over RCCL — and rank 0 prints the line.  Under a launcher (the driver's torch.distributed.run) it reads RANK / LOCAL_RANK /
  WORLD_SIZE / MASTER_* as given.

--workload silesia (default; BASELINE.json configs[1], compress of the same blocks = configs[2] under "compress"):
  the Silesia stand-in `silesia_mix` (211 938 580 B, rust-lz-fear_amd/synth.py) cut into 4 MiB independent blocks and tiled
  `--copies` times per GPU so that one launch has enough independent blocks to occupy 256 CUs.  `--distinct` copies are
  generated with their own seeds ("tiled with distinct seeds"); copy k beyond them is distinct copy k % distinct
  rotated by a copy-specific number of bytes (the 4 MiB cuts fall elsewhere, so the blocks differ in content and cost) and
  XOR-ed with a copy-specific byte.  A "step" is one decompress pass over every block this rank owns: the kernel over the
  compressed blocks + the device copy the frame layer does for stored blocks (framed/decompress.rs:250).  Inputs are resident
  in HBM before the timed region.  Blocks shard across ranks with no data-path collective ("weak": per-GPU work fixed).

--workload config4 (BASELINE.json configs[3]): a `log_text` stream of 2048 x 4 MiB blocks (8 GiB; --blocks scales it), framed
  independent-blocks mode, sharded by contiguous block ranges across the ranks.  A step = device compress of the rank's
  blocks -> size table + payload all-gather over torch.distributed (RCCL) -> the frame assembled in HBM on every rank, all
  inside the timed region ("strong": total work fixed).  The frame is checked against the oracle's frame on a prefix.
"""
import argparse
import ctypes as C
import json
import multiprocessing
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import synth  # noqa: E402

BS = 4 << 20
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "GiB/s compress + decompress, 4 MiB independent blocks, 1/2/4/8 MI355X"


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


# --------------------------------------------------------------------------------------- corpus (host, before CUDA is touched)
def _gen_copy(args):
    k, path = args
    np.save(path, synth.silesia_mix(copy=k))
    return path


def make_distinct_copies(distinct):
    """`distinct` copies of silesia_mix with their own seeds, generated in parallel (numpy, one process each)."""
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="lzfbench_", dir=shm)
    jobs = [(k, os.path.join(tmp, f"c{k}.npy")) for k in range(distinct)]
    nproc = max(1, min(distinct, (os.cpu_count() or 2) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    if nproc > 1:
        with multiprocessing.get_context("fork").Pool(nproc) as pool:
            paths = pool.map(_gen_copy, jobs)
    else:
        paths = [_gen_copy(j) for j in jobs]
    arrs = [np.load(p) for p in paths]
    for p in paths:
        os.unlink(p)
    os.rmdir(tmp)
    return arrs


# --------------------------------------------------------------------------------------- CPU baseline (native threads)
def build_cpubench():
    """oracle/lzf_cpu_bench.c + lzf_oracle.c compiled HERE for this host (-march=native), into a temporary directory."""
    d = tempfile.mkdtemp(prefix="lzfcpu_")
    so = os.path.join(d, "liblzf_cpubench.so")
    src = [os.path.join(ROOT, "oracle", f) for f in ("lzf_cpu_bench.c", "lzf_oracle.c")]
    subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=gnu11", "-pthread", "-shared", "-o", so] + src + ["-ldl"])
    L = C.CDLL(so)
    L.lzfo_bench_run.restype = C.c_int
    L.lzfo_bench_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.lzfo_bench_liblz4_version.restype = C.c_char_p
    return L


def cpu_model():
    model, gov = "unknown", "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        gov = open("/sys/devices/system/cpu/cpu0/cpufreq/scaling_governor").read().strip()
    except OSError:
        gov = "n/a (no cpufreq in this guest)"
    return model, gov


def cpu_baseline(raw_blocks, comp_blocks, budget_s):
    """The CPU restatement of lz-fear (oracle/, kind "port": the Rust reference cannot be built here) and liblz4 (the C
    implementation the reference compares itself with, README.md:11,18), timed with native threads on this host:
    single thread and all hardware threads, median of 5 passes each, over the same 4 MiB blocks the GPU decodes."""
    L = build_cpubench()
    cores = os.cpu_count() or 1
    n = len(raw_blocks)
    nbytes = sum(len(b) for b in raw_blocks)
    reps = 5

    def arrs(blocks, mult):
        ptrs = (C.c_void_p * (n * mult))(*([b.ctypes.data for b in blocks] * mult))
        lens = (C.c_uint64 * (n * mult))(*([len(b) for b in blocks] * mult))
        return ptrs, lens

    def run(what, threads, mult):
        rp, rl = arrs(raw_blocks, mult)
        cp, cl = arrs(comp_blocks, mult)
        secs = (C.c_double * reps)()
        inp, inl = (rp, rl) if what in (0, 2) else (cp, cl)
        rc = L.lzfo_bench_run(what, inp, inl, rl, n * mult, threads, reps, secs)
        if rc != 0:
            return None
        return round(nbytes * mult / sorted(secs)[reps // 2] / 2**30, 3)

    # enough block-jobs per pass that every thread gets several (the same blocks again: read-only inputs)
    mult_all = max(1, (4 * cores + n - 1) // n)
    t0 = time.time()
    out = {"single_thread": {"decompress": run(1, 1, 1), "compress": run(0, 1, 1)},
           "all_cores": {"decompress": run(1, cores, mult_all), "compress": run(0, cores, mult_all)}}
    ver = L.lzfo_bench_liblz4_version()
    if ver and time.time() - t0 < budget_s:
        out["liblz4"] = {"version": ver.decode(), "single_thread": {"decompress": run(3, 1, 1), "compress": run(2, 1, 1)},
                         "all_cores": {"decompress": run(3, cores, mult_all), "compress": run(2, cores, mult_all)}}
    else:
        out["liblz4"] = None
    model, gov = cpu_model()
    out.update({"value": out["all_cores"]["decompress"], "unit": "GiB/s (decompress, uncompressed bytes; all hardware threads)",
                "cores": cores, "kind": "port", "cpu_model": model, "governor": gov, "reps": reps,
                "compress_value": out["all_cores"]["compress"],
                "sample": f"the {n} compressible 4 MiB blocks of one corpus copy ({nbytes / 2**20:.0f} MiB), {mult_all} x per all-core pass; "
                          f"native pthreads (oracle/lzf_cpu_bench.c, gcc -O3 -march=native), median of {reps} passes; "
                          "oracle/lzf_oracle.c = lz-fear restated in C; liblz4 = LZ4_compress_fast_continue on a fresh stream / LZ4_decompress_safe"})
    return out


# --------------------------------------------------------------------------------------- HBM traffic (PMC passes under profiles/)
def traffic_for(kernel_name, which, n_jobs):
    """HBM bytes per call of the named path: FETCH_SIZE (x2: gfx950 correction) + WRITE_SIZE from the separate --pmc passes committed
    under profiles/ (tools/refresh_profiles.sh), summed over every kernel the call launches.  `which` = "decompress" (the headline
    call: the pass runs at 48 copies — counter collection at 240 does not finish — and is SCALED by job count), "tile20" (the pass
    runs at exactly this size: measured, not scaled) or "compress".  The second value says which; (None, None) when the committed
    profile is of other kernels than the ones this run launched."""
    tj = path = None
    for name in ("r06_hbm_traffic.json", "r05_hbm_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            tj = json.load(open(path)).get(which)
            if tj:
                break
    if not tj:
        return None, None
    head = lambda k: k.split("<")[0].split(" +")[0].split(":")[0].strip()        # noqa: E731
    if head(tj.get("kernel", "")) != head(kernel_name):
        return None, None
    scale = float(n_jobs) / float(tj["jobs"])
    measured = abs(scale - 1.0) < 0.02
    info = {"kind": "measured at this size by a separate PMC pass" if measured else "scaled by job count from a separate PMC pass at a smaller size",
            "pass_jobs": int(tj["jobs"]), "this_run_jobs": int(n_jobs), "profile": os.path.basename(path),
            "formula": "2 x FETCH_SIZE + WRITE_SIZE over every kernel of the call (gfx950: FETCH_SIZE counts half of wide streaming reads)"}
    return (2 * tj["FETCH_SIZE_KB"] + tj["WRITE_SIZE_KB"]) * 1024.0 * scale, info


def last_decompress_launch(ffi):
    """What the last lzf_decompress_batch of this thread launched, as the library says (lzf_last_decompress_launch)."""
    return ffi.lib().lzf_last_decompress_launch().decode()


def issue_ceiling(copies, kernel_ms, achieved_gbs, cus=256, which="decompress"):
    """What the call retires against what a dependent SALU + VALU mix reaches on this chip (tools/issue_mix_microbench.hip, r01: 3.29
    wave-instructions / ns / CU at 32 waves per CU).  Wave-instructions per sequence from the round's PMC pass (all kernels of the call)."""
    ipseq, seq_per_copy, best_rate = 27.96, 11.71e6, 3.29
    kind = "model (constants from earlier counter passes; only kernel_ms is this run's)"
    for name in ("r06_issue_counters.json", "r05_issue_counters.json"):
        pc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pc):
            m = json.load(open(pc)).get(which)
            if m:
                ipseq = float(m["wave_instructions_per_sequence"])
                kind = "wave-instructions per sequence from the PMC pass profiles/%s (%s), kernel_ms from this run, best_rate from tools/issue_mix_microbench.hip (r01)" % (
                    name, ", ".join(f"{k[9:]} {v}" for k, v in sorted(m["per_sequence"].items()) if k.startswith("SQ_INSTS")))
                break
    rate = ipseq * seq_per_copy * copies / (kernel_ms * 1e-3) / 1e9 / cus
    ceil_gbs = achieved_gbs * best_rate / rate
    return {"kind": kind, "wave_instructions_per_sequence": ipseq, "retired_per_ns_per_cu": round(rate, 3), "best_measured_mix_per_ns_per_cu": best_rate,
            "ceiling_gbs": round(ceil_gbs, 1), "ceiling_frac_of_hbm": round(ceil_gbs / HBM_PEAK_GBS, 4), "achieved_frac_of_ceiling": round(rate / best_rate, 3)}


def timed_launches(torch, fn, steps):
    """Run fn() `steps` times; HIP events on the launch stream bracket each call."""
    evs = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    return evs


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: the same command line under torch.distributed.run, one rank per GPU."""
    # --standalone: the launcher's own rendezvous picks the port (no bind-then-close race between concurrent runs on one box)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}",
           os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] --gpus {n} without a launcher: " + " ".join(cmd))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def launch_check(args, rank, world, emit):
    """The launcher path without a GPU: gloo group, every rank's block range of the config4 stream gathered, rank 0 prints them."""
    import torch
    import torch.distributed as dist
    from rust_lz_fear_amd.dist import shard_range
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    lo, hi = shard_range(args.blocks, rank, world)
    mine = torch.tensor([rank, lo, hi], dtype=torch.int64)
    allr = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allr, mine)
    dist.barrier()
    if rank == 0:
        emit({"launch_check": True, "world": dist.get_world_size(), "blocks": args.blocks, "ranges": [[int(x) for x in t.tolist()] for t in allr]})
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["silesia", "config4", "config5"], default="silesia")
    ap.add_argument("--copies", type=int, default=240, help="silesia: tiled copies of silesia_mix per GPU (51 blocks each)")
    ap.add_argument("--distinct", type=int, default=12, help="silesia: copies generated with their own seeds (the rest are rotated + XOR-ed)")
    ap.add_argument("--blocks", type=int, default=2048, help="config4: 4 MiB blocks of the whole stream (2048 = 8 GiB)")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the CPU-baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-config4", action="store_true", help="silesia workload: skip the strong-scaled configs[3] leg of the line")
    ap.add_argument("--no-config5", action="store_true", help="silesia workload: skip the configs[4] leg of the line (linked 64 KiB blocks + dictionary, U16Table)")
    ap.add_argument("--no-sweep", action="store_true", help="silesia workload: skip the batch-size sweep and the tile20 leg (PMC passes: the headline call's kernels only)")
    ap.add_argument("--streams", type=int, default=4096, help="config5: linked-block streams of 1 MiB per GPU")
    ap.add_argument("--launch-check", action="store_true",
                    help="no GPU work: every rank joins a gloo group, the ranks agree on their block ranges, rank 0 prints them (test of the self-launch path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: become one (the ranks inherit stdout: rank 0's JSON line is this process's output)
        sys.exit(self_launch(args.gpus))
    # stdout carries ONE JSON line and nothing else: libraries that write to the C-level stdout (RCCL prints a version banner when a
    # communicator is made) are sent to stderr; the line goes to the real stdout through its saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.launch_check:
        return launch_check(args, rank, world, emit)

    t0 = time.time()
    bases = None
    if args.workload == "silesia":
        bases = make_distinct_copies(max(1, min(args.distinct, args.copies)))
        log(f"[bench] {len(bases)} distinct-seed copies of silesia_mix generated ({time.time() - t0:.1f}s)")

    import torch
    from rust_lz_fear_amd import device, ffi
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:     # under torch.distributed.run: always the RCCL path
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", device_id=dev)
        dist = dist_mod
    assert ffi.device_count() >= 1

    if args.workload == "config4":
        line = run_config4(args, torch, device, ffi, dist, rank, world, dev)
    elif args.workload == "config5":
        line = run_config5(args, torch, device, ffi, dist, rank, world, dev)
    else:
        line = run_silesia(args, torch, device, ffi, dist, rank, world, dev, bases)
        # The two legs below run BEHIND the Silesia measurement and must never take its line down with them (a first multi-GPU run
        # executes code no one-GPU box ever has): a failure becomes {"error": ...} under the leg's key, on every rank alike.
        def leg(name, fn, timeout_s=900.0):
            import threading
            import traceback

            def bail():                     # a leg that hangs (a collective one rank never enters) cannot be interrupted from Python:
                if rank == 0 and line is not None:      # the watchdog emits the Silesia line without it and ends the process
                    line[name] = {"error": f"the {name} leg did not return within {timeout_s:.0f} s; the line was emitted without it"}
                    emit(line)
                os._exit(0)
            wd = threading.Timer(timeout_s, bail); wd.daemon = True; wd.start()
            ok, out = 1, None
            try:
                out = fn()
            except BaseException as e:      # noqa: BLE001
                ok, out = 0, {"error": f"{type(e).__name__}: {e}", "where": traceback.format_exc()[-1500:]}
                log(f"[bench] rank {rank}: the {name} leg failed: {out['error']}")
            if dist:                        # every rank must leave the leg (its collectives) before the next one starts
                try:
                    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    if int(flag[0]) == 0 and ok:
                        out = {"error": "another rank failed in this leg"}
                except BaseException as e:  # noqa: BLE001
                    out = {"error": f"{name}: the ranks could not agree on the outcome: {e}"}
            wd.cancel()
            return out
        if not args.no_config4:
            # the one workload with a real exchange (BASELINE configs[3]), strong-scaled, as a leg of the same line: a driver
            # run at N GPUs records the weak-scaled Silesia value and this
            import copy, gc

            def c4():
                gc.collect(); torch.cuda.empty_cache()
                a4 = copy.copy(args); a4.steps = max(1, min(args.steps, 3)); a4.warmup = 1; a4.no_cpu = True
                l4 = run_config4(a4, torch, device, ffi, dist, rank, world, dev)
                if rank != 0:
                    return None
                return {"value": l4["value"], "unit": l4["unit"], "scaling": "strong", "steps": a4.steps, "ms_per_step": l4["ms_per_step"],
                        "compress_ms": l4["roofline"]["kernel_ms"], "gather_ms": l4["config"]["gather_ms"],
                        "wire_bytes_in_per_rank": l4["config"]["wire_bytes_in_per_rank"], "frame_bytes": l4["config"]["frame_bytes"],
                        "n_ranks_seen_by_rccl": l4["config"]["n_ranks_seen_by_rccl"], "gather_path": l4["config"]["gather_path"], "blocks": l4["config"]["blocks"],
                        "librccl": l4["config"]["librccl"],
                        "compress_launch": l4["roofline"]["kernel"],
                        "verified_against_oracle_prefix": l4["config"]["verified_against_oracle_prefix"],
                        "content_checksum": l4["config"]["content_checksum"]}
            r4 = leg("config4", c4)
            if rank == 0:
                line["config4"] = r4
        if not args.no_config5 and world == 1:
            # configs[4]: linked 64 KiB blocks behind a dictionary (many streams in lock-step) and raw U16Table jobs, device-resident
            import copy, gc

            def c5():
                gc.collect(); torch.cuda.empty_cache()
                a5 = copy.copy(args); a5.steps = max(1, min(args.steps, 3)); a5.warmup = 1
                l5 = run_config5(a5, torch, device, ffi, dist, rank, world, dev)
                return {k: l5[k] for k in ("value", "unit", "steps", "ms_per_step", "config", "roofline", "compress", "u16_raw")} if rank == 0 else None
            r5 = leg("config5", c5)
            if rank == 0:
                line["config5"] = r5
    if rank == 0:
        emit(line)
    if dist:
        dist.destroy_process_group()


# ======================================================================================= configs[1] / configs[2]
def run_silesia(args, torch, device, ffi, dist, rank, world, dev, bases):
    t0 = time.time()
    total = bases[0].size
    nb1 = (total + BS - 1) // BS                      # 51 blocks per copy, the last one short
    copies = args.copies
    nblk = nb1 * copies
    d_bases = [torch.from_numpy(b).to(dev) for b in bases]
    nd = len(d_bases)
    src = torch.zeros(nblk * BS, dtype=torch.uint8, device=dev)
    lens = np.full(nblk, BS, dtype=np.uint64)
    for k in range(copies):
        g = rank * copies + k                                         # global copy index: no two copies anywhere hold the same bytes
        b = d_bases[g % nd]
        gen = g // nd                                                 # 0: the distinct copy itself
        dst = src[k * nb1 * BS:k * nb1 * BS + total]
        if gen == 0:
            dst.copy_(b)
        else:
            shift = (gen * 1000003 + (g % nd) * 65537) % total        # rotate: the 4 MiB cuts land elsewhere
            c = (gen * 37 + (gen >> 3) + 1) & 0xFF
            torch.bitwise_xor(torch.roll(b, shift), c, out=dst)
        lens[k * nb1 + nb1 - 1] = total - (nb1 - 1) * BS
    del d_bases
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
        device.compress_batch(d_cj, d_cres, nblk, ffi.KINDS_U32 | ffi.KINDS_U32_FRESH_ONLY)

    c_steps = max(1, min(args.steps, 3))
    compress_step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    tc0 = time.perf_counter()
    c_evs = timed_launches(torch, compress_step, c_steps)
    c_launch = ffi.lib().lzf_last_compress_launch().decode()
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
    # how evenly the launch fills the chip: the jobs' own wave times (results[].reserved, kilo-cycles) against slots x duration
    c_kc = cres["reserved"].astype(np.float64) * 1024.0
    c_fill = {"sum_of_wave_gcycles": round(float(c_kc.sum()) / 1e9, 2), "longest_job_mcycles": round(float(c_kc.max()) / 1e6, 1),
              "mean_job_mcycles": round(float(c_kc.mean()) / 1e6, 1)}
    c_bytes = float(lens.sum() + clen.sum())                           # N + C

    # stored blocks (incompressible, framed/compress.rs:250-255): the frame carries the raw bytes
    stored_idx = np.nonzero(full)[0]
    comp2d, src2d = comp.view(nblk, BS), src.view(nblk, BS)
    t_idx = None
    if len(stored_idx):
        t_idx = torch.from_numpy(stored_idx.astype(np.int64)).to(dev)
        comp2d[t_idx] = src2d[t_idx]
        # device-side move of the stored blocks during decompression (lzf_copy_ranges, on a second stream beside the kernel)
        sidx = stored_idx.astype(np.uint64)
        d_sp = torch.from_numpy((np.uint64(comp.data_ptr()) + sidx * np.uint64(BS)).view(np.int64)).to(dev)
        d_sl = torch.from_numpy(lens[stored_idx].astype(np.uint64).view(np.int64)).to(dev)
        side = torch.cuda.Stream(device=dev)

    # ------------------------------------------------------------------ decompress (configs[1])
    dec = torch.zeros(nblk * BS, dtype=torch.uint8, device=dev)
    kidx = np.nonzero(ok)[0]
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
        d_dp = torch.from_numpy((np.uint64(dec.data_ptr()) + sidx * np.uint64(BS)).view(np.int64)).to(dev)
    evs = []

    def decompress_step():
        if t_idx is not None:                                          # framed/decompress.rs:250, next to the kernel
            side.wait_stream(torch.cuda.current_stream())
            device.copy_ranges(d_sp, d_dp, d_sl, len(stored_idx), BS, stream=side)
        evs.extend(timed_launches(torch, decompress_kernel, 1))
        if t_idx is not None:
            torch.cuda.current_stream().wait_stream(side)

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
    kname_main = last_decompress_launch(ffi)

    dres = device.results_to_host(d_dres, nk)
    assert np.all(dres["status"] == ffi.OK), f"decompress statuses: {np.unique(dres['status'])}"
    assert np.array_equal(dres["out_len"], lens[kidx])
    if not args.no_verify:
        assert torch.equal(dec, src), "decoded bytes differ from the source"   # round trip at full size

    t = torch.tensor([elapsed, tc], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(lens.sum()), float(lens[kidx].sum())], dtype=torch.float64, device=dev)
    if dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed_max, tc_max = float(t[0]), float(t[1])
    total_bytes, kernel_bytes = float(tot[0]), float(tot[1])

    d_bytes = float(lens[kidx].sum() + clen[kidx].sum())               # C + N of the kernel's jobs
    kname = kname_main
    d_traffic, d_traffic_info = traffic_for(kname, "decompress", nk)
    c_traffic, c_traffic_info = traffic_for(c_launch, "compress", nblk)
    d_achieved = d_bytes / (d_kernel_ms * 1e-3) / 1e9
    c_achieved = c_bytes / (c_kernel_ms * 1e-3) / 1e9

    # ------------------------------------------------------------------ batch-size sweep of the same call (this rank's first copies)
    sweep = tile20 = None
    if rank == 0 and not args.no_sweep:
        sweep = {}
        per_copy = max(1, nk // copies)
        for c_n in (1, 4, 20):
            m = min(nk, per_copy * c_n)
            if m == nk:
                continue
            ts = []
            for _ in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); device.decompress_batch(d_dj, d_dres, m); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            r_ = device.results_to_host(d_dres, m)
            assert np.all(r_["status"] == ffi.OK) and np.array_equal(r_["out_len"], lens[kidx[:m]])
            ms = float(sorted(ts[1:])[2])
            sweep[str(m)] = {"ms": round(ms, 3), "gibs": round(float(lens[kidx[:m]].sum()) / (ms * 1e-3) / 2**30, 2), "launch": last_decompress_launch(ffi)}
            if c_n == 20:
                # BASELINE.md's planned throughput tile ("tiled x20 with distinct seeds, 1020 blocks") as a leg of its own:
                # the same call over the first 20 copies' compressed blocks, with its own roofline (HBM traffic MEASURED at this size)
                t_bytes = float(lens[kidx[:m]].sum() + clen[kidx[:m]].sum())
                t_ach = t_bytes / (ms * 1e-3) / 1e9
                t_traffic, t_info = traffic_for(sweep[str(m)]["launch"], "tile20", m)
                tile20 = {"value": sweep[str(m)]["gibs"], "unit": "GiB/s (uncompressed bytes decompressed per second, one call over 20 copies)", "blocks": int(m),
                          "copies": 20, "ms_per_step": round(ms, 3),
                          "roofline": {"bound": "hbm", "kernel": sweep[str(m)]["launch"], "achieved": round(t_ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(t_ach / HBM_PEAK_GBS, 5), "traffic": t_traffic, "traffic_provenance": t_info,
                                       "algorithmic_bytes_per_launch": t_bytes, "kernel_ms": round(ms, 3),
                                       "north_star_frac": round(float(lens[kidx[:m]].sum()) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}}
        sweep[str(nk)] = {"ms": round(d_kernel_ms, 3), "gibs": round(float(lens[kidx].sum()) / (d_kernel_ms * 1e-3) / 2**30, 2), "launch": kname_main}
        if not args.no_verify:
            assert torch.equal(dec, src), "decoded bytes differ from the source after the batch sweep"

    # ------------------------------------------------------------------ CPU baseline + host-buffer end to end (rank 0, N = 1)
    cpu = e2e = None
    if rank == 0 and world == 1:
        first = [i for i in range(nb1) if ok[i]]
        raw_blocks = [bases[0][i * BS:i * BS + int(lens[i])] for i in first]
        # (the host-buffer legs first: with the CPU baseline or even just its input copies in front of them, the same calls
        #  measured 50-58 ms instead of 42-47 on the same box)
        if not args.no_e2e:
            del dec
            e2e = end_to_end(bases, ffi)
        if not args.no_cpu:
            comp_blocks = [comp2d[i, : int(clen[i])].cpu().numpy() for i in first]
            cpu = cpu_baseline(raw_blocks, comp_blocks, args.cpu_seconds)

    if rank != 0:
        return None
    value = total_bytes * args.steps / elapsed_max / 2**30
    return {
        "metric": METRIC,
        "value": round(value, 3), "unit": "GiB/s (uncompressed bytes decompressed per second)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "silesia_mix (Silesia stand-in, 211938580 B) x %d copies per GPU (%d with their own seeds, the rest "
                               "rotated + XOR-ed), 4 MiB independent blocks, decompress_raw of every block (configs[1]); lz4 ratio %.3f" %
                               (copies, len(bases), float(lens.sum()) / float(clen.sum() + lens[stored_idx].sum())),
                   "blocks_per_gpu": int(nblk), "stored_blocks_per_gpu": int(len(stored_idx)),
                   "block_size": BS, "parallelism": f"block-sharded x{world}, no collective",
                   "cu_count": int(torch.cuda.get_device_properties(dev).multi_processor_count)},     # what the library's dispatch thresholds are derived from
        # the kernel alone over the compressed blocks (the stored ones are a device memcpy beside it)
        "kernel_only": {"value": round(kernel_bytes / world / (d_kernel_ms * 1e-3) / 2**30 * world, 3), "unit": "GiB/s over the compressed blocks only",
                        "blocks_per_gpu": int(nk)},
        "roofline": {"bound": "hbm", "kernel": kname,
                     "achieved": round(d_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(d_achieved / HBM_PEAK_GBS, 5), "traffic": d_traffic, "traffic_provenance": d_traffic_info,
                     "algorithmic_bytes_per_launch": d_bytes, "kernel_ms": round(d_kernel_ms, 4),
                     "north_star_frac": round(float(lens[kidx].sum()) / (d_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     # what one workgroup per block can reach at its instruction count: SQ counters of the launched kernel
                     # (profiles/r02_decompress_generations.txt: 9.0 SALU + 16.1 VALU + 2.6 LDS + 0.24 VMEM per sequence; 11.71 M sequences
                     # per corpus copy) against the issue rate a dependent SALU + VALU mix reaches (profiles/r01_issue_mix_microbench.txt)
                     "issue_ceiling": issue_ceiling(copies, d_kernel_ms, d_achieved)},
        "compress": {"value": round(total_bytes * c_steps / tc_max / 2**30, 3), "unit": "GiB/s (uncompressed bytes compressed per second)",
                     "steps": c_steps, "ms_per_step": round(tc_max / c_steps * 1e3, 3),
                     "roofline": {"bound": "hbm", "kernel": c_launch,
                                  "achieved": round(c_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(c_achieved / HBM_PEAK_GBS, 5), "traffic": c_traffic, "traffic_provenance": c_traffic_info,
                                  "algorithmic_bytes_per_launch": c_bytes, "kernel_ms": round(c_kernel_ms, 3),
                                  "issue_ceiling": issue_ceiling(copies, c_kernel_ms, c_achieved, which="compress")},
                     "launch_fill": c_fill},
        # the same call at smaller batch sizes (compressed blocks of the first 1 / 4 / 20 copies; kernel time by HIP events,
        # median of 5): up to 1024 blocks go through the segmented pipeline, a block decoded by many wavefronts
        "batch_sweep": sweep,
        # the x20 tile of BASELINE.md / SURVEY §8(d) (1 020 blocks, 980 of them compressed): its own value and roofline
        "tile20": tile20,
        "cpu_baseline": cpu,
        "end_to_end": e2e,
    }


def end_to_end(bases, ffi):
    """The C ABI from HOST buffers (lzfear_frame.h): frames of 16 MiB, independent blocks, content checksum — at the reference's
    default block size (4 MiB) and at 64 KiB blocks, 64 frames (1 GiB) per call; and, at 4 MiB blocks, the same calls at 4 GiB and
    16 GiB per call (`call_size_sweep`: 256 blocks per call is exactly the one-block-per-CU latency class of both kernels, the larger
    calls are the throughput class).  PCIe in and out, frame scan / assembly and checksums included — never the bench `value`."""
    from rust_lz_fear_amd import framed
    L = ffi.lib()
    F, fsz = 64, 16 << 20
    mix = bases[0]
    distinct = [mix[(i * fsz) % (mix.size - fsz):][:fsz].tobytes() for i in range(F)]
    res = {"frames": F, "frame_bytes": fsz, "settings": "independent blocks, content checksum (CompressionSettings::default())",
           "buffers": "pageable host memory", "calls": "median of 5 after one warm-up call (the first call pins the staging slab); the 4 and 16 GiB calls: median of 3 after one"}

    def point(bs, n, reps):
        datas = [distinct[i % F] for i in range(n)]                    # (inputs may repeat: every frame still has its own output buffer)
        total = n * fsz
        s = framed.CompressionSettings().block_size(bs)._struct(None)
        cap = L.lzf_frame_compress_bound(C.byref(s), fsz)
        outs = [C.create_string_buffer(cap) for _ in range(n)]
        ins = (C.c_char_p * n)(*datas)
        lens = (C.c_size_t * n)(*[fsz] * n)
        outp = (C.c_void_p * n)(*[C.addressof(o) for o in outs])
        capa = (C.c_size_t * n)(*[cap] * n)
        olen = (C.c_size_t * n)()
        st = (C.c_int * n)()
        tcs = []
        for _ in range(reps + 1):
            t = time.perf_counter()
            rc = L.lzf_frame_compress_many(C.byref(s), n, ins, lens, outp, capa, olen, st)
            tcs.append(time.perf_counter() - t)
            assert rc == 0 and not any(st)
        frames = [C.string_at(outs[f], olen[f]) for f in range(n)]
        del outs
        dcap = fsz + 64
        douts = [C.create_string_buffer(dcap) for _ in range(n)]
        fin = (C.c_char_p * n)(*frames)
        flen = (C.c_size_t * n)(*[len(f) for f in frames])
        doutp = (C.c_void_p * n)(*[C.addressof(o) for o in douts])
        dcapa = (C.c_size_t * n)(*[dcap] * n)
        dlen = (C.c_size_t * n)()
        used = (C.c_size_t * n)()
        dst = (C.c_int * n)()
        tds = []
        for _ in range(reps + 1):
            t = time.perf_counter()
            rc = L.lzf_frame_decompress_many(n, fin, flen, None, 0, doutp, dcapa, dlen, used, dst)
            tds.append(time.perf_counter() - t)
            assert rc == 0 and not any(dst)
        assert all(C.string_at(douts[f], dlen[f]) == datas[f] for f in range(0, n, max(8, n // 16)))
        mc, md = sorted(tcs[1:])[len(tcs[1:]) // 2], sorted(tds[1:])[len(tds[1:]) // 2]
        return {"frame_compress_many_gibs": round(total / mc / 2**30, 3), "frame_decompress_many_gibs": round(total / md / 2**30, 3),
                "frame_compress_many_ms": round(mc * 1e3, 1), "frame_decompress_many_ms": round(md * 1e3, 1)}

    for bs, key in ((4 << 20, "4MiB_blocks"), (64 << 10, "64KiB_blocks")):
        res[key] = point(bs, F, 5)
    sweep = {"1GiB": dict(res["4MiB_blocks"], blocks=F * 4)}
    try:
        avail = int([l.split()[1] for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0]) * 1024
    except Exception:
        avail = 0
    for gib in (4, 16):
        n = gib * 64
        if avail < 4 * gib * 2**30:                                    # frames + both sets of output buffers + the staging's own
            sweep[f"{gib}GiB"] = {"skipped": f"{avail >> 30} GiB of host memory available"}
            continue
        try:
            sweep[f"{gib}GiB"] = dict(point(4 << 20, n, 3), blocks=n * 4)
        except Exception as e:                                          # (never the line: this leg is extra)
            sweep[f"{gib}GiB"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    res["call_size_sweep"] = {"what": "4 MiB blocks, frames of 16 MiB, one lzf_frame_compress_many / lzf_frame_decompress_many call over 1 / 4 / 16 GiB of host memory", **sweep}
    L.lzf_frame_release_scratch()
    return res


# ======================================================================================= configs[3]
def run_config4(args, torch, device, ffi, dist, rank, world, dev):
    """8 GiB log-text stream, framed independent-blocks mode, block ranges sharded over the ranks, frame reassembled by all-gather."""
    from rust_lz_fear_amd import dist as lzdist, framed
    nblk_all = args.blocks
    lo, hi = lzdist.shard_range(nblk_all, rank, world)
    nloc = hi - lo
    t0 = time.time()
    # synthetic stream: 64 MiB of log_text generated, tile t of it rotated by a tile-specific number of bytes and its digits
    # re-keyed (XOR with a tile byte on the low nibble of digit positions would break the text; a plain rotation keeps it log text)
    base_blocks = 16
    base = torch.from_numpy(synth.log_text(0, base_blocks * BS)).to(dev)
    src = torch.empty(nloc * BS, dtype=torch.uint8, device=dev)
    for i in range(nloc):
        g = lo + i
        tile, b = divmod(g, base_blocks)
        blk = base[b * BS:(b + 1) * BS]
        src[i * BS:(i + 1) * BS] = torch.roll(blk, (tile * 104729) % BS) if tile else blk
    torch.cuda.synchronize()
    log(f"[bench] rank {rank}: blocks [{lo}, {hi}) of {nblk_all} ({src.numel() / 2**30:.2f} GiB) in HBM ({time.time() - t0:.1f}s)")
    comp = torch.empty(nloc * BS, dtype=torch.uint8, device=dev)
    cj = np.zeros(nloc, dtype=device.CJOB)
    cj["input"] = np.uint64(src.data_ptr()) + np.arange(nloc, dtype=np.uint64) * np.uint64(BS)
    cj["input_len"] = BS
    cj["out"] = np.uint64(comp.data_ptr()) + np.arange(nloc, dtype=np.uint64) * np.uint64(BS)
    cj["out_cap"] = BS
    cj["table_kind"] = ffi.TABLE_U32
    d_cj = device.to_device(cj, dev)
    d_cres = torch.zeros(nloc * 16, dtype=torch.uint8, device=dev)
    frame_cap = 64 + nblk_all * (BS + 8)
    frame = torch.empty(min(frame_cap, 64 + nblk_all * 8 + (nblk_all * BS) // 2), dtype=torch.uint8, device=dev)   # log text compresses > 2 x
    state = {}
    header = lzdist.frame_header(content_checksum=False, block_size=BS)

    # the exchange under the C ABI (include/lzfear_dist.h: lzf_frame_gather over an RCCL communicator of its own, bootstrapped by
    # broadcasting rank 0's ncclUniqueId over the process group); torch.distributed's P2P path stays as the fallback
    comm, gather_path = None, "torch.distributed (all_gather_into_tensor + batch_isend_irecv)"
    if os.environ.get("LZF_GATHER", "c") != "torch":
        try:
            comm = lzdist.DistComm(dist, rank, world, dev)
            gather_path = "lzf_frame_gather (C ABI, liblzfear_dist.so: ncclAllGather + grouped ncclSend / ncclRecv)"
        except Exception as e:      # noqa: BLE001 — a first multi-rank run must not die in new code
            log(f"[bench] rank {rank}: lzf_dist_comm_init failed ({e}); falling back to torch.distributed")
            gather_path += f" [fallback: {e}]"
    if dist and world > 1:      # every rank takes the same path
        flag = torch.tensor([1 if comm else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0 and comm:
            comm.close(); comm = None
            gather_path = "torch.distributed (all_gather_into_tensor + batch_isend_irecv) [fallback: another rank could not make the communicator]"

    # which librccl the exchange library is bound to, next to every copy mapped into this process (torch carries its own)
    try:
        librccl = lzdist.rccl_paths()
    except Exception as e:      # noqa: BLE001
        librccl = {"error": str(e)}
    # a first exchange outside the timed region: if the C path fails at STEP time on any rank (lzf_frame_gather returns the same code on
    # every rank for what one rank can see, include/lzfear_dist.h), every rank moves to the torch path together
    if comm:
        ok = 1
        try:
            device.compress_batch(d_cj, d_cres, nloc, ffi.KINDS_U32 | ffi.KINDS_U32_FRESH_ONLY)
            lzdist.gather_frame_device_c(comm, d_cres, comp, src, BS, nloc, nblk_all, frame, header)
        except Exception as e:      # noqa: BLE001
            ok = 0
            log(f"[bench] rank {rank}: lzf_frame_gather failed at step time ({e}); falling back to torch.distributed")
            gather_path = f"torch.distributed (all_gather_into_tensor + batch_isend_irecv) [fallback: lzf_frame_gather failed: {e}]"
        if dist and world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 0 and ok:
                ok = 0
                gather_path = "torch.distributed (all_gather_into_tensor + batch_isend_irecv) [fallback: lzf_frame_gather failed on another rank]"
        if not ok:
            comm.close(); comm = None

    kev, gev = [], []

    def step():
        a, b, c2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        device.compress_batch(d_cj, d_cres, nloc, ffi.KINDS_U32 | ffi.KINDS_U32_FRESH_ONLY)
        state["launch"] = ffi.lib().lzf_last_compress_launch().decode()
        b.record()
        kev.append((a, b))
        if comm:
            state["frame_len"], state["comp_total"] = lzdist.gather_frame_device_c(comm, d_cres, comp, src, BS, nloc, nblk_all, frame, header)
        else:
            state["frame_len"], state["comp_total"] = lzdist.gather_frame_device(d_cres, comp, src, BS, nloc, nblk_all, frame, dist, rank, world, device, header)
        c2.record()
        gev.append((b, c2))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    kev.clear(); gev.clear()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t[0])
    k_ms = sum(a.elapsed_time(b) for a, b in kev) / max(len(kev), 1)      # the compress call of this rank (cost probe + kernel), HIP events
    g_ms = sum(a.elapsed_time(b) for a, b in gev) / max(len(gev), 1)      # size-table all-gather + packing + segment exchange + header / EndMark
    # what the reference's default (content_checksum: true) would add: XXH32 does not compose across ranks, so it is one serial
    # chain over the whole stream on one host thread — timed here on a 256 MiB sample, outside the timed region
    xx = None
    if rank == 0:
        sample = src[: min(src.numel(), 256 << 20)].cpu().numpy().tobytes()
        tx = time.perf_counter(); ffi.lib().lzf_xxh32(sample, len(sample), 0); tx = time.perf_counter() - tx
        xx = {"in_timed_region": False, "host_xxh32_gb_per_s": round(len(sample) / tx / 1e9, 2),
              "serial_ms_for_the_stream": round(float(nblk_all) * BS / (len(sample) / tx) * 1e3, 1)}
    # ---- check: the frame's head == the oracle's frame of the same first blocks (rank 0 holds them)
    verified = None
    if rank == 0 and not args.no_verify:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_ffi as o
        k = min(3, nloc)
        host = src[: k * BS].cpu().numpy().tobytes()
        rc, ref = o.frame_compress(host, settings=o.make_settings(content_checksum=False))
        assert rc == 0
        mine = frame[: len(ref)].cpu().numpy().tobytes()
        body = ref[:-4]                                                # (without the oracle frame's EndMark)
        verified = mine[: len(body)] == body
        assert verified, "sharded frame differs from the oracle's frame on the first blocks"
    n_seen = comm.count() if comm else (dist.get_world_size() if dist else 0)
    if comm:
        comm.close()
    if rank != 0:
        return None
    total_bytes = float(nblk_all) * BS
    wire = state["comp_total"] * (world - 1) / max(world, 1)
    # roofline of the dominant kernel (this rank's launch): reads N, writes C
    alg = float(nloc) * BS + state["comp_total"] * nloc / max(nblk_all, 1)
    achieved = alg / (k_ms * 1e-3) / 1e9
    cpu = None
    if world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        n_s = min(nloc, 16)
        raw_blocks = [src[i * BS:(i + 1) * BS].cpu().numpy() for i in range(n_s)]
        res = d_cres.view(torch.int64).view(-1, 2)[:n_s].cpu().numpy()
        comp_blocks = [comp[i * BS: i * BS + int(res[i, 0])].cpu().numpy() for i in range(n_s)]
        cpu = cpu_baseline(raw_blocks, comp_blocks, args.cpu_seconds)
        cpu["value"], cpu["unit"] = cpu["compress_value"], "GiB/s (compress, uncompressed bytes; all hardware threads)"
    return {
        "metric": METRIC,
        "value": round(total_bytes * args.steps / elapsed_max / 2**30, 3), "unit": "GiB/s (uncompressed bytes compressed into one frame per second)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "config4: log_text stream of %d x 4 MiB blocks (%.1f GiB), framed independent-blocks mode (content checksum "
                               "off: XXH32 of the whole stream is one serial chain), blocks sharded by contiguous ranges, size table + "
                               "payload all-gather and frame assembly inside the timed region; lz4 ratio %.2f" %
                               (nblk_all, total_bytes / 2**30, total_bytes / max(state["comp_total"], 1)),
                   "blocks": nblk_all, "block_size": BS, "parallelism": f"block ranges x{world}, all-gather over RCCL",
                   "frame_bytes": int(state["frame_len"]), "wire_bytes_in_per_rank": int(wire), "verified_against_oracle_prefix": verified,
                   "gather_ms": round(g_ms, 3), "gather_path": gather_path,
                   "n_ranks_seen_by_rccl": n_seen, "librccl": librccl, "content_checksum": xx},
        "roofline": {"bound": "hbm", "kernel": state["launch"], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic_for(state["launch"], "compress", nloc)[0],
                     "algorithmic_bytes_per_launch": alg, "kernel_ms": round(k_ms, 4)},
        "cpu_baseline": cpu,
    }


# ======================================================================================= configs[4]
def run_config5(args, torch, device, ffi, dist, rank, world, dev):
    """BASELINE configs[4], device-resident: S streams of 1 MiB — each a 256-byte motif of its own, repeated — as frames of 64 KiB LINKED
    blocks behind a 64 KiB dictionary of the same motif (the U32Table path, the only one the reference's frame layer has,
    framed/compress.rs:202-214,:271-275; framed/decompress.rs:238-269): block k of every stream in launch k, the table and the window
    carried on the device (lzf_table_offset_batch / lzf_chain_decompress_step).  Plus the raw U16Table path (mod.rs:78-101) on
    65 535-byte slices of the same data.  Weak-scaled like the headline (every rank its own streams, no collective)."""
    B5, SL, DL = 64 << 10, 1 << 20, 64 << 10
    S = args.streams
    NB = SL // B5
    stride = DL + SL
    t0 = time.time()
    g = torch.Generator(device="cpu"); g.manual_seed(0x5EED0005 + rank)
    motifs = torch.randint(0, 256, (S, 256), dtype=torch.uint8, generator=g).to(dev)
    slab = motifs.repeat(1, stride // 256).contiguous().view(-1)               # stream s = slab[s * stride + DL : (s + 1) * stride], its dictionary in front of it
    del motifs
    TSZ = 4096 * 4 + 8                                                            # sizeof(lzf_u32_table)
    d_tmpl = torch.zeros(S * TSZ, dtype=torch.uint8, device=dev)
    d_tabs = torch.zeros(S * TSZ, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib = ffi.lib()
    torch.cuda.synchronize(); ts = time.perf_counter()
    for s_ in range(S):                                                           # template_table (framed/compress.rs:202-211), one per dictionary
        ffi.check(lib.lzf_table_seed_from_dictionary(d_tmpl.data_ptr() + s_ * TSZ, slab.data_ptr() + s_ * stride, DL, st))
    torch.cuda.synchronize(); seed_ms = (time.perf_counter() - ts) * 1e3
    comp = torch.empty(S * NB * B5, dtype=torch.uint8, device=dev)              # slot stride = block size (cap = N)
    nj = S * NB
    sidx = np.arange(S, dtype=np.uint64)
    cj = np.zeros(nj, dtype=device.CJOB)
    for k in range(NB):                                                            # step-major: jobs of step k are [k * S, (k + 1) * S)
        v = cj[k * S:(k + 1) * S]
        v["input"] = np.uint64(slab.data_ptr()) + sidx * np.uint64(stride) + np.uint64(k * B5)    # in_buffer = the last 64 KiB ++ block (:217-222)
        v["input_len"] = DL + B5
        v["cursor"] = DL
        v["out"] = np.uint64(comp.data_ptr()) + (sidx * np.uint64(NB) + np.uint64(k)) * np.uint64(B5)
        v["out_cap"] = B5                                                          # framed/compress.rs:242
        v["table"] = np.uint64(d_tabs.data_ptr()) + sidx * np.uint64(TSZ)
    cj["table_kind"] = ffi.TABLE_U32
    d_cj = device.to_device(cj, dev)
    d_cres = torch.zeros(nj * 16, dtype=torch.uint8, device=dev)
    d_tabptr = torch.from_numpy((np.uint64(d_tabs.data_ptr()) + sidx * np.uint64(TSZ)).view(np.int64)).to(dev)
    d_adds = torch.full((S,), B5, dtype=torch.int64, device=dev)                  # table.offset(forget): the window forgets 64 KiB per block (:271-275)
    log(f"[bench] rank {rank}: config5: {S} streams x {SL >> 20} MiB ({S * SL / 2**30:.2f} GiB) + dictionaries in HBM, templates seeded in {seed_ms:.0f} ms ({time.time() - t0:.1f}s)")

    def compress_step():
        d_tabs.copy_(d_tmpl)                                                       # table = template.clone() (:213-214)
        for k in range(NB):
            if k:
                ffi.check(lib.lzf_table_offset_batch(d_tabptr.data_ptr(), d_adds.data_ptr(), S, ffi.TABLE_U32, st))
            ffi.check(lib.lzf_compress_batch(d_cj.data_ptr() + k * S * 56, d_cres.data_ptr() + k * S * 16, S, ffi.KINDS_U32, st))

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        evs = []
        t1 = time.perf_counter()
        for _ in range(steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        if dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0]), float(np.mean([a.elapsed_time(b) for a, b in evs]))

    c_el, c_ms = timed(compress_step, args.steps, args.warmup)
    c_launch = lib.lzf_last_compress_launch().decode()
    cres = device.results_to_host(d_cres, nj)
    assert np.all(cres["status"] == ffi.OK), f"config5 compress statuses: {np.unique(cres['status'])}"
    clen = cres["out_len"].astype(np.uint64)

    # ---- decompress: the stream's output is one buffer, the window is what it already holds (framed/decompress.rs:253-269)
    dstride = SL + 2 * B5
    dec = torch.zeros(S * dstride, dtype=torch.uint8, device=dev)
    dj = np.zeros(nj, dtype=device.DJOB)
    CS = np.dtype([("prev_job", "<u4"), ("job", "<u4"), ("stored_len", "<u8"), ("stored_src", "<u8"), ("out", "<u8"), ("block_maxsize", "<u8")])
    cs = np.zeros(nj, dtype=CS)
    for k in range(NB):
        v = dj[k * S:(k + 1) * S]
        v["input"] = cj["out"][k * S:(k + 1) * S]
        v["input_len"] = clen[k * S:(k + 1) * S]
        v["prefix"] = np.uint64(slab.data_ptr()) + sidx * np.uint64(stride)      # the dictionary (:239-245)
        v["prefix_len"] = DL
        v["out"] = np.uint64(dec.data_ptr()) + sidx * np.uint64(dstride)
        v["out_cap"] = B5 + clen[k * S:(k + 1) * S]                                # (patched per step by lzf_chain_decompress_step)
        v["output_limit"] = B5
        c = cs[k * S:(k + 1) * S]
        c["prev_job"] = (np.uint32(0xFFFFFFFF) if k == 0 else (np.arange(S, dtype=np.uint32) + np.uint32((k - 1) * S)))
        c["job"] = np.arange(S, dtype=np.uint32) + np.uint32(k * S)
        c["out"] = v["out"]
        c["block_maxsize"] = B5
    d_dj = device.to_device(dj, dev)
    d_cs = device.to_device(cs, dev)
    d_dres = torch.zeros(nj * 16, dtype=torch.uint8, device=dev)
    d_state = torch.zeros(S * 16, dtype=torch.uint8, device=dev)

    def decompress_step():
        d_state.zero_()
        for k in range(NB):
            ffi.check(lib.lzf_chain_decompress_step(d_cs.data_ptr() + k * S * 40, d_state.data_ptr(), S, d_dj.data_ptr(), d_dres.data_ptr(), st))
            ffi.check(lib.lzf_decompress_batch_sized(d_dj.data_ptr() + k * S * 64, d_dres.data_ptr() + k * S * 16, S, B5, st))

    d_el, d_ms = timed(decompress_step, args.steps, args.warmup)
    d_launch = lib.lzf_last_decompress_launch().decode()
    dres = device.results_to_host(d_dres, nj)
    assert np.all(dres["status"] == ffi.OK), f"config5 decompress statuses: {np.unique(dres['status'])}"
    if not args.no_verify:
        assert torch.equal(dec.view(S, dstride)[:, :SL], slab.view(S, stride)[:, DL:]), "config5: decoded streams differ from the source"
    verified = None
    if rank == 0 and not args.no_verify:
        # the first and the last stream against the oracle's linked-block loop (framed/compress.rs:221-276), block by block
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ctypes as C
        import oracle_ffi as o
        for s_ in (0, S - 1):
            host = slab[s_ * stride:(s_ + 1) * stride].cpu().numpy().tobytes()
            dic, data = host[:DL], host[DL:]
            tab = o.new_table()
            contract = C.c_int(0)
            rep = o.lib().lzfo_u32_replace
            rep.restype = C.c_size_t
            rep.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
            for off in range(0, len(dic) - 7, 3):
                rep(C.addressof(tab), dic, len(dic), off, C.byref(contract))
            buf = dic
            for k in range(NB):
                blk = data[k * B5:(k + 1) * B5]
                inp = buf + blk
                erc, ecomp = o.compress2(inp, cursor=len(buf), table=tab, cap=len(blk))
                q = k * S + s_
                mine = comp[(s_ * NB + k) * B5:(s_ * NB + k) * B5 + int(clen[q])].cpu().numpy().tobytes()
                assert erc == 0 and mine == ecomp, f"config5: stream {s_} block {k} differs from the oracle"
                buf = inp
                if len(buf) > 65536:
                    forget = len(buf) - 65536
                    tab.offset += forget
                    buf = buf[forget:]
        verified = True

    # ---- raw U16Table jobs (mod.rs:78-101): 65 535-byte slices, fresh tables; and back
    NU = min(4 * S, (S * stride) // 65536)
    uj = np.zeros(NU, dtype=device.CJOB)
    uoff = np.arange(NU, dtype=np.uint64) * np.uint64(65536)
    uj["input"] = np.uint64(slab.data_ptr()) + uoff
    uj["input_len"] = 65535
    uj["out"] = np.uint64(comp.data_ptr()) + uoff
    uj["out_cap"] = 65535
    uj["table_kind"] = ffi.TABLE_U16
    d_uj = device.to_device(uj, dev)
    d_ures = torch.zeros(NU * 16, dtype=torch.uint8, device=dev)
    u_el, u_ms = timed(lambda: ffi.check(lib.lzf_compress_batch(d_uj.data_ptr(), d_ures.data_ptr(), NU, ffi.KINDS_U16, st)), args.steps, args.warmup)
    ures = device.results_to_host(d_ures, NU)
    assert np.all(ures["status"] == ffi.OK)
    ud = np.zeros(NU, dtype=device.DJOB)
    ud["input"] = uj["out"]; ud["input_len"] = ures["out_len"]
    ud["out"] = np.uint64(dec.data_ptr()) + uoff; ud["out_cap"] = 65535 + 64; ud["output_limit"] = 65535
    d_ud = device.to_device(ud, dev)
    d_udres = torch.zeros(NU * 16, dtype=torch.uint8, device=dev)
    dec.zero_()
    ud_el, ud_ms = timed(lambda: ffi.check(lib.lzf_decompress_batch_sized(d_ud.data_ptr(), d_udres.data_ptr(), NU, 65536, st)), args.steps, args.warmup)
    udres = device.results_to_host(d_udres, NU)
    assert np.all(udres["status"] == ffi.OK) and np.all(udres["out_len"] == 65535)
    if not args.no_verify:
        assert torch.equal(dec[: NU * 65536].view(NU, 65536)[:, :65535], slab[: NU * 65536].view(NU, 65536)[:, :65535]), "config5: U16 round trip differs"
    if rank == 0 and not args.no_verify:
        import oracle_ffi as o
        h = slab[:65535].cpu().numpy().tobytes()
        erc, ecomp = o.compress2(h, kind=o.TABLE_U16)
        assert erc == 0 and comp[: int(ures["out_len"][0])].cpu().numpy().tobytes() == ecomp, "config5: U16Table job differs from the oracle"
    if rank != 0:
        return None
    tot = float(S) * SL * world
    N1, C1 = float(S) * SL, float(clen.sum())
    gib = lambda bytes_, secs: round(bytes_ / secs / 2**30, 3)
    roof = lambda alg, ms, kern: {"bound": "hbm", "kernel": kern, "achieved": round(alg / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes_per_launch": alg, "kernel_ms": round(ms, 3),
                                  "launches_per_step": None}
    r_d = roof(N1 + C1, d_ms, d_launch); r_d["launches_per_step"] = f"{NB} x (lzf_chain_decompress_step + lzf_decompress_batch of {S} jobs)"
    r_c = roof(N1 + C1, c_ms, c_launch); r_c["launches_per_step"] = f"table clone + {NB} x (lzf_table_offset_batch + lzf_compress_batch of {S} jobs)"
    return {
        "metric": METRIC, "value": gib(tot * args.steps, d_el), "unit": "GiB/s (uncompressed bytes decompressed per second; linked 64 KiB blocks behind a dictionary)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(d_el / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "config5: %d streams x 1 MiB per GPU, each a 256-byte motif of its own repeated, framed as 64 KiB LINKED blocks behind a "
                               "64 KiB dictionary of the motif (U32Table, table and window carried on the device, block k of every stream in launch k); "
                               "lz4 ratio %.0f" % (S, N1 / max(C1, 1.0)),
                   "streams_per_gpu": S, "block_size": B5, "blocks_per_stream": NB, "dictionary_bytes": DL, "template_seed_ms_untimed": round(seed_ms, 1),
                   "verified_against_oracle": verified, "parallelism": f"streams x{world}, no collective (linked streams do not shard: replicas only)"},
        "roofline": r_d,
        "compress": {"value": gib(tot * args.steps, c_el), "unit": "GiB/s (uncompressed bytes compressed per second)", "ms_per_step": round(c_el / args.steps * 1e3, 3), "roofline": r_c},
        "u16_raw": {"jobs": int(NU), "slice_bytes": 65535,
                    "compress": {"value": gib(float(NU) * 65535 * world * args.steps, u_el), "unit": "GiB/s", "ms_per_step": round(u_el / args.steps * 1e3, 3),
                                 "roofline": roof(float(NU) * 65535 + float(ures["out_len"].sum()), u_ms, "lzf_compress_wave_kernel<U16>")},
                    "decompress": {"value": gib(float(NU) * 65535 * world * args.steps, ud_el), "unit": "GiB/s", "ms_per_step": round(ud_el / args.steps * 1e3, 3),
                                   "roofline": roof(float(NU) * 65535 + float(ures["out_len"].sum()), ud_ms, lib.lzf_last_decompress_launch().decode())}},
    }


if __name__ == "__main__":
    main()
