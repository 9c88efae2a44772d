#!/bin/bash
# Regenerates the round's measurement artifacts on the GPU box (run through gpurun from the repo root), product library only:
#   gpurun_out/r04/bench_default.json          python bench.py (the driver's default invocation)
#   gpurun_out/r04/kernel_stats.txt            rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/r04/pmc_{FETCH,WRITE}_SIZE.txt  separate --pmc passes at 48 copies (1 step; 2352 compressed blocks: the same kernels
#                                               as the default run — counter collection at 240 copies does not finish) for HBM traffic
#   gpurun_out/r04/hbm_traffic.json            the two passes as the file bench.py reads (copy to profiles/r04_hbm_traffic.json)
#   gpurun_out/r04/bench_config4.json          python bench.py --workload config4
#   gpurun_out/r04/bench_240x48.json, bench_240x1.json   seed-sensitivity check at full size: 240 copies from 48 seeds / from 1 seed
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; rm -rf $O; mkdir -p $O
cd $R && timeout 1500 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1500 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --no-cpu --no-e2e --no-config4 > $O/prof.log 2>&1)
DB=$(ls $O/prof/*/x_results.db $O/prof/x_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $O/kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $R && timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --copies 48 --distinct 1 --steps 1 --warmup 0 --no-cpu --no-e2e --no-verify --no-config4 > $O/pmc_$c.log 2>&1)
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
                "(48 copies: the same kernels as the default 240-copy run; bench.py scales by job count) on MI355X, round 4; per dispatch, in the counters' KB units (x1024 bytes). "
                "gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of wide streaming reads; bench.py uses 2 x FETCH + WRITE as the upper bound."}
dk = line["roofline"]["kernel"].replace(",", ", ")        # the launched variant exactly, as rocprofv3 prints it (the batch sweep launches others)
for which, needle, jobs in (("decompress", dk, line["kernel_only"]["blocks_per_gpu"]), ("compress", "lzf_compress_compact_kernel<false>", line["config"]["blocks_per_gpu"])):
    f, w = per_dispatch("FETCH_SIZE", needle), per_dispatch("WRITE_SIZE", needle)
    if f and w:
        out[which] = {"kernel": line["roofline"]["kernel"] if which == "decompress" else "lzf_compress_compact_kernel<false>", "kernel_as_profiled": f[0], "grid": f[2],
                      "jobs": jobs, "FETCH_SIZE_KB": f[1], "WRITE_SIZE_KB": w[1]}
json.dump(out, open(os.path.join(O, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
cd $R && timeout 900 python bench.py --workload config4 > $O/bench_config4.log 2>&1; tail -1 $O/bench_config4.log > $O/bench_config4.json
# the segmented pipeline, kernel by kernel, at 1 / 4 / 8 / 15 copies of the corpus (49 ... 735 blocks), and the pair kernel on the same batches
bash $R/tools/gpu_seg_stats.sh 1 4 8 15 > $O/seg_kernel_stats.txt 2>&1
VARIANT=noseg bash $R/tools/gpu_seg_stats.sh 1 4 8 15 > $O/noseg_kernel_stats.txt 2>&1
cut -c1-1500 $O/bench_default.json; head -14 $O/kernel_stats.txt; cat $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt; cut -c1-600 $O/bench_config4.json; cat $O/seg_kernel_stats.txt; grep "==\|jobs\|paired" $O/noseg_kernel_stats.txt
