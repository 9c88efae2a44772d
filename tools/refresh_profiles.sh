#!/bin/bash
# Regenerates the round's measurement artifacts on the GPU box (run through gpurun from the repo root), product library only:
#   gpurun_out/r06/bench_default.json          python bench.py (the driver's default invocation)
#   gpurun_out/r06/kernel_stats.txt            rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/r06/pmc_{FETCH,WRITE}_SIZE.txt  separate --pmc passes at 48 copies (1 step; 2352 compressed blocks: the same kernels
#                                               as the default run — counter collection at 240 copies does not finish) for HBM traffic
#   gpurun_out/r06/hbm_traffic.json            the two passes as the file bench.py reads (copy to profiles/r06_hbm_traffic.json)
#   gpurun_out/r06/bench_config4.json          python bench.py --workload config4
#   gpurun_out/r06/issue_counters.json         SQ instruction counters of both headline kernels per sequence (copy to profiles/r06_issue_counters.json)
#   gpurun_out/r06/bench_config5.json          python bench.py --workload config5
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; rm -rf $O; mkdir -p $O
cd $R && timeout 1500 python bench.py > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1500 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --no-cpu --no-e2e --no-config4 --no-config5 > $O/prof.log 2>&1)
DB=$(ls $O/prof/*/x_results.db $O/prof/x_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $O/kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $R && timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --copies 48 --distinct 1 --steps 1 --warmup 0 --no-cpu --no-e2e --no-verify --no-config4 --no-config5 > $O/pmc_$c.log 2>&1)
  rm -rf $O/pmc_$c/*/*.db
  f=$(ls $O/pmc_$c/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" "$c" > $O/pmc_$c.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter(); grid = {}
for r in csv.DictReader(open(sys.argv[1])):
    if 'lzf' in r['Kernel_Name']:
        k = r['Kernel_Name'][:90]; agg[k] += float(r['Counter_Value']); cnt[k] += 1; grid[k] = r.get('Grid_Size', '')
for k in agg: print(sys.argv[2], '|', k, '| dispatches', cnt[k], '| grid', grid[k], '| sum', agg[k], '| per_dispatch', agg[k] / cnt[k])
PY
done
python - $O <<'PY'
import json, os, re, sys
O = sys.argv[1]
line = json.loads([l for l in open(os.path.join(O, "pmc_FETCH_SIZE.log")) if l.startswith('{"metric"')][-1])     # the profiled run's own line: its kernels and job counts
def per_dispatch(counter, needle):
    best = None
    for l in open(os.path.join(O, f"pmc_{counter}.txt")):
        p = [x.strip() for x in l.split("|")]
        if needle in p[1]:
            best = (p[1], float(p[5].split()[-1]), p[3].split()[-1])
    return best
out = {"_what": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace) of `python bench.py --copies 48 --distinct 1 --steps 1 --warmup 0 --no-cpu --no-e2e --no-verify` "
                "(48 copies: the same kernels as the default 240-copy run; bench.py scales by job count) on MI355X, round 6; per dispatch, in the counters' KB units (x1024 bytes). "
                "gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of wide streaming reads; bench.py uses 2 x FETCH + WRITE as the upper bound."}
dk = line["roofline"]["kernel"].replace(",", ", ")        # the launched variant exactly, as rocprofv3 prints it (the batch sweep launches others)
for which, needle, jobs in (("decompress", dk, line["kernel_only"]["blocks_per_gpu"]), ("compress", "lzf_compress_compact_kernel<false>", line["config"]["blocks_per_gpu"])):
    f, w = per_dispatch("FETCH_SIZE", needle), per_dispatch("WRITE_SIZE", needle)
    if f and w:
        out[which] = {"kernel": line["roofline"]["kernel"] if which == "decompress" else line["compress"]["roofline"]["kernel"], "kernel_as_profiled": f[0], "grid": f[2],
                      "jobs": jobs, "FETCH_SIZE_KB": f[1], "WRITE_SIZE_KB": w[1]}
json.dump(out, open(os.path.join(O, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
# ---- instruction counters of the two headline kernels (same 48-copy command): what roofline.issue_ceiling is computed from
(cd $R && timeout 420 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $O/pmc_insts -- python bench.py --copies 48 --distinct 1 --steps 1 --warmup 0 --no-cpu --no-e2e --no-verify --no-config4 --no-config5 > $O/pmc_insts.log 2>&1)
rm -rf $O/pmc_insts/*/*.db
python - $O <<'PY'
import csv, glob, json, os, sys, collections
O = sys.argv[1]
f = glob.glob(os.path.join(O, "pmc_insts", "*", "*_counter_collection.csv"))[0]
line = json.loads([l for l in open(os.path.join(O, "pmc_insts.log")) if l.startswith('{"metric"')][-1])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "lzf" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); cnt[k] += 1
SEQ_PER_COPY = 11.71e6          # sequences of one copy of the corpus (oracle statistics, tools/seq_stats.c)
out = {"_what": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES (one pass, with --kernel-trace) of "
                "`python bench.py --copies 48 --distinct 1 --steps 1 --warmup 0 ...` on MI355X, round 6; wave-instructions per dispatch and per sequence "
                "(48 copies x 11.71 M sequences)."}
dk = line["roofline"]["kernel"].replace(",", ", ")
for which, needle in (("decompress", dk), ("compress", "lzf_compress_compact_kernel<false>")):
    ks = [k for k in agg if needle in k]
    if not ks: continue
    k = ks[0]; n = cnt[k]; c = {m: v / n for m, v in agg[k].items()}
    seqs = SEQ_PER_COPY * 48
    tot = sum(c.get(m, 0) for m in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
    out[which] = {"kernel_as_profiled": k[:120], "dispatches": n, "per_dispatch": c, "sequences": seqs,
                  "per_sequence": {m: round(v / seqs, 2) for m, v in c.items()}, "wave_instructions_per_sequence": round(tot / seqs, 2)}
json.dump(out, open(os.path.join(O, "issue_counters.json"), "w"), indent=1)
print(json.dumps({k: v.get("per_sequence") for k, v in out.items() if isinstance(v, dict)}, indent=1))
PY
cd $R && timeout 900 python bench.py --workload config4 > $O/bench_config4.log 2>&1; grep '^{"metric"' $O/bench_config4.log | tail -1 > $O/bench_config4.json
cd $R && timeout 900 python bench.py --workload config5 > $O/bench_config5.log 2>&1; grep '^{"metric"' $O/bench_config5.log | tail -1 > $O/bench_config5.json
cut -c1-2500 $O/bench_default.json; head -14 $O/kernel_stats.txt; cat $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt; cut -c1-600 $O/bench_config4.json; cut -c1-400 $O/bench_config5.json
