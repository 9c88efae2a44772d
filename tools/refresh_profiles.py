#!/usr/bin/env python3
"""Regenerates the round's measurement artifacts on the GPU box (run through gpurun from the repo root), product library only.

    gpurun_out/<round>/bench_default.json        python bench.py (the driver's default invocation)
    gpurun_out/<round>/kernel_stats.txt          rocprofv3 --kernel-trace --stats of the same command (profiles/summarize_rocpd.py)
    gpurun_out/<round>/pmc_<size>_<pass>.txt     separate --pmc passes (FETCH_SIZE | WRITE_SIZE | the SQ instruction counters), each
                                                 with --kernel-trace only, of `bench.py --copies C --distinct 1 --steps 1 --warmup 0
                                                 --no-sweep ...`: C = 48 (the headline call's kernels: counter collection at 240
                                                 copies does not finish; bench.py scales by job count) and C = 20 (the tile20 leg:
                                                 measured at its own size)
    gpurun_out/<round>/hbm_traffic.json          the FETCH / WRITE passes as the file bench.py's traffic_for() reads
    gpurun_out/<round>/issue_counters.json       wave-instructions per sequence, the call as a whole and kernel by kernel
    gpurun_out/<round>/bench_config4.json, bench_config5.json

A decompress CALL is several kernels since round 6 (hop parse + seam + bitmap-fed copy stage + the pair kernel over what is left,
or the segmented pipeline's stages): every counter is summed over all kernels of the call, and listed per kernel beside the sum.
Copy the files you want judged to profiles/<round>_*.

usage: python tools/refresh_profiles.py [round-tag, default r06] [--quick: skip config4/5 and the default bench line]"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
TAG = next((a for a in sys.argv[1:] if not a.startswith("-")), "r06")
QUICK = "--quick" in sys.argv
FULL = "--full-traffic" in sys.argv          # also try the FETCH / WRITE passes at the benchmark's own 240 copies (earlier rounds: did not finish)
O = os.path.join(R, "gpurun_out", TAG)
os.makedirs(O, exist_ok=True)
ENV = dict(os.environ, TMPDIR="/tmp")
SEQ_PER_COPY = 11.71e6          # sequences of one copy of the corpus (oracle statistics, tools/seq_stats.c)
COMMON = ["--no-cpu", "--no-e2e", "--no-config4", "--no-config5"]
SQ = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES"]


def sh(cmd, log, timeout, cwd=R):
    with open(log, "w") as f:
        try:
            return subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, cwd=cwd, env=ENV, timeout=timeout).returncode
        except subprocess.TimeoutExpired:
            f.write("\n[refresh] timed out\n")
            return 124


def json_line(log):
    ls = [l for l in open(log, errors="replace") if l.startswith('{"metric"')]
    return json.loads(ls[-1]) if ls else None


def is_decompress_kernel(k):
    return "lzf" in k and not any(s in k for s in ("lzf_compress", "lzf_copy_ranges", "lzf_cost_probe", "lzf_order_by_cost", "lzf_xxh32"))


def short(k):
    k = k.replace("void ", "").replace("lzf::", "")
    return k.split("(")[0]


def pmc_pass(size, name, counters, timeout=600):
    """One rocprofv3 --pmc pass of the bench at `size` copies.  -> ({kernel: {counter: sum}}, {kernel: dispatches}, the run's line)"""
    d = os.path.join(O, f"pmc_{size}_{name}")
    log = d + ".log"
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "--", "python", "bench.py", "--copies", str(size), "--distinct", "1",
           "--steps", "1", "--warmup", "0", "--no-verify", "--no-sweep", *COMMON]
    rc = sh(cmd, log, timeout)
    for db in glob.glob(os.path.join(d, "*", "*.db")):
        os.remove(db)
    fs = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    if fs:
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"]
            if "lzf" not in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
    with open(os.path.join(O, f"pmc_{size}_{name}.txt"), "w") as f:
        f.write(f"# rc {rc}; rocprofv3 --kernel-trace --pmc {' '.join(counters)} -- python bench.py --copies {size} --distinct 1 --steps 1 --warmup 0 --no-verify --no-sweep {' '.join(COMMON)}\n")
        for k in sorted(agg):
            f.write(f"{short(k)[:80]} | dispatches {len(cnt[k])} | " + " | ".join(f"{c} {v:.6g}" for c, v in sorted(agg[k].items())) + "\n")
    return agg, {k: len(v) for k, v in cnt.items()}, json_line(log)


def main():
    if not QUICK:
        sh(["python", "bench.py"], os.path.join(O, "bench_default.log"), 1500)
        l = json_line(os.path.join(O, "bench_default.log"))
        if l:
            json.dump(l, open(os.path.join(O, "bench_default.json"), "w"))
        prof = os.path.join(O, "prof")
        subprocess.run(["rm", "-rf", prof])
        sh(["rocprofv3", "--kernel-trace", "--stats", "-d", prof, "-o", "x", "--", "python", "bench.py", *COMMON], os.path.join(O, "prof.log"), 1500)
        dbs = glob.glob(os.path.join(prof, "*", "x_results.db")) + glob.glob(os.path.join(prof, "x_results.db"))
        if dbs:
            with open(os.path.join(O, "kernel_stats.txt"), "w") as f:
                subprocess.run(["python", os.path.join(R, "profiles", "summarize_rocpd.py"), dbs[0]], stdout=f)
            for db in dbs:
                os.remove(db)

    traffic = {"_what": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only) of `python bench.py --copies C --distinct 1 --steps 1 "
                        "--warmup 0 --no-verify --no-sweep ...` on MI355X, round 6: C = 48 for `decompress` and `compress` (the same kernels as the default 240-copy run; "
                        "bench.py scales by job count), C = 20 for `tile20` (measured at its own size).  Per CALL: summed over every kernel of the call (`per_kernel` lists them), "
                        "in the counters' KB units (x1024 bytes).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of wide streaming "
                        "reads; bench.py uses 2 x FETCH + WRITE as the upper bound."}
    issue = {"_what": "rocprofv3 --pmc " + " ".join(SQ) + " (one pass, with --kernel-trace only) of the same commands; wave-instructions per call and per sequence "
                      "(C copies x 11.71 M sequences), summed over every kernel of the call and kernel by kernel."}
    for size, key in ((48, "decompress"), (20, "tile20")):
        passes = {}
        for name, counters in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("insts", SQ)):
            passes[name] = pmc_pass(size, name, counters)
        line = passes["FETCH_SIZE"][2] or passes["WRITE_SIZE"][2] or passes["insts"][2]
        if not line:
            continue
        launch = line["roofline"]["kernel"]
        jobs = line["kernel_only"]["blocks_per_gpu"]
        fa, wa = passes["FETCH_SIZE"][0], passes["WRITE_SIZE"][0]
        ks = sorted(k for k in set(fa) | set(wa) if is_decompress_kernel(k))
        per_kernel = {short(k): {"dispatches": passes["FETCH_SIZE"][1].get(k, 0), "FETCH_SIZE_KB": fa[k].get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB": wa[k].get("WRITE_SIZE", 0.0)} for k in ks}
        traffic[key] = {"kernel": launch, "jobs": jobs, "copies": size, "calls": 1,
                        "FETCH_SIZE_KB": sum(v["FETCH_SIZE_KB"] for v in per_kernel.values()), "WRITE_SIZE_KB": sum(v["WRITE_SIZE_KB"] for v in per_kernel.values()),
                        "per_kernel": per_kernel}
        ia = passes["insts"][0]
        seqs = SEQ_PER_COPY * size
        tot = collections.defaultdict(float)
        pk = {}
        for k in sorted(ia):
            if not is_decompress_kernel(k):
                continue
            c = dict(ia[k])
            for m, v in c.items():
                tot[m] += v
            n = sum(c.get(m, 0.0) for m in SQ[:5])
            pk[short(k)] = {"dispatches": passes["insts"][1].get(k, 0), "wave_instructions_per_sequence": round(n / seqs, 3),
                            "per_sequence": {m: round(v / seqs, 3) for m, v in c.items()}}
        issue[key] = {"kernel": launch, "copies": size, "sequences": seqs, "per_call": dict(tot),
                      "per_sequence": {m: round(v / seqs, 2) for m, v in tot.items()},
                      "wave_instructions_per_sequence": round(sum(tot.get(m, 0.0) for m in SQ[:5]) / seqs, 2), "per_kernel": pk}
        if size == 48:
            ck = [k for k in fa if "lzf_compress_compact_kernel<false>" in k]
            if ck:
                k = ck[0]
                n = max(1, passes["FETCH_SIZE"][1].get(k, 1))
                traffic["compress"] = {"kernel": line["compress"]["roofline"]["kernel"], "kernel_as_profiled": short(k), "jobs": line["config"]["blocks_per_gpu"], "copies": size,
                                       "calls": n, "FETCH_SIZE_KB": fa[k]["FETCH_SIZE"] / n, "WRITE_SIZE_KB": wa[k]["WRITE_SIZE"] / max(1, passes["WRITE_SIZE"][1].get(k, 1))}
                c = {m: v / max(1, passes["insts"][1].get(k, 1)) for m, v in ia[k].items()}
                issue["compress"] = {"kernel_as_profiled": short(k), "dispatches": passes["insts"][1].get(k, 0), "per_dispatch": c, "sequences": seqs,
                                     "per_sequence": {m: round(v / seqs, 2) for m, v in c.items()},
                                     "wave_instructions_per_sequence": round(sum(c.get(m, 0.0) for m in SQ[:5]) / seqs, 2)}
    if FULL and "decompress" in traffic:
        # the headline call at its own size: only the two traffic passes (each runs the 240-copy bench once under the counters)
        fa, n1, line = pmc_pass(240, "FETCH_SIZE", ["FETCH_SIZE"], timeout=1500)
        wa, n2, line2 = pmc_pass(240, "WRITE_SIZE", ["WRITE_SIZE"], timeout=1500)
        line = line or line2
        ks = sorted(k for k in set(fa) | set(wa) if is_decompress_kernel(k))
        if line and ks and all(k in fa and k in wa for k in ks if "fed_kernel" in k or "parse" in k):
            per_kernel = {short(k): {"dispatches": n1.get(k, 0), "FETCH_SIZE_KB": fa[k].get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB": wa[k].get("WRITE_SIZE", 0.0)} for k in ks}
            traffic["decompress_48"] = traffic["decompress"]
            traffic["decompress"] = {"kernel": line["roofline"]["kernel"], "jobs": line["kernel_only"]["blocks_per_gpu"], "copies": 240, "calls": 1,
                                     "FETCH_SIZE_KB": sum(v["FETCH_SIZE_KB"] for v in per_kernel.values()), "WRITE_SIZE_KB": sum(v["WRITE_SIZE_KB"] for v in per_kernel.values()),
                                     "per_kernel": per_kernel}
            ck = [k for k in fa if "lzf_compress_compact_kernel<false>" in k]
            if ck and ck[0] in wa:
                k = ck[0]
                traffic["compress_48"] = traffic.get("compress")
                traffic["compress"] = {"kernel": line["compress"]["roofline"]["kernel"], "kernel_as_profiled": short(k), "jobs": line["config"]["blocks_per_gpu"], "copies": 240,
                                       "calls": max(1, n1.get(k, 1)), "FETCH_SIZE_KB": fa[k]["FETCH_SIZE"] / max(1, n1.get(k, 1)), "WRITE_SIZE_KB": wa[k]["WRITE_SIZE"] / max(1, n2.get(k, 1))}
    json.dump(traffic, open(os.path.join(O, "hbm_traffic.json"), "w"), indent=1)
    json.dump(issue, open(os.path.join(O, "issue_counters.json"), "w"), indent=1)
    for key in ("decompress", "tile20", "compress"):
        t, i = traffic.get(key), issue.get(key)
        if t:
            print(key, "| jobs", t["jobs"], "| 2 x FETCH + WRITE =", round((2 * t["FETCH_SIZE_KB"] + t["WRITE_SIZE_KB"]) * 1024 / 1e9, 2), "GB |", t["kernel"][:100])
        if i:
            print("   wave-instructions per sequence:", i["wave_instructions_per_sequence"], {k: v["wave_instructions_per_sequence"] for k, v in i.get("per_kernel", {}).items() if v["wave_instructions_per_sequence"] >= 0.01})

    if not QUICK:
        for w in ("config4", "config5"):
            sh(["python", "bench.py", "--workload", w], os.path.join(O, f"bench_{w}.log"), 900)
            l = json_line(os.path.join(O, f"bench_{w}.log"))
            if l:
                json.dump(l, open(os.path.join(O, f"bench_{w}.json"), "w"))
        for f in ("bench_default.json",):
            p = os.path.join(O, f)
            if os.path.exists(p):
                print(open(p).read()[:2500])
        p = os.path.join(O, "kernel_stats.txt")
        if os.path.exists(p):
            print("".join(open(p).readlines()[:14]))


if __name__ == "__main__":
    main()
