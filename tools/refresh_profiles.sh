# Regenerates the round's measurement artifacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/r01/bench_default.json          python bench.py (the driver's default invocation)
#   gpurun_out/r01/kernel_stats.txt            rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/r01/pmc_{fetch,write}.csv       separate --pmc passes (40 copies, 1 step, the kernel the full-size bench
#                                               selects: paired24) for HBM traffic
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01; rm -rf $O; mkdir -p $O
cd $R && timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py > $O/prof.log 2>&1)
DB=$(ls $O/prof/*/x_results.db $O/prof/x_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $O/kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $R && LZF_DECOMPRESS_KERNEL=paired24 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --copies 40 --steps 1 --warmup 0 --no-cpu --no-verify > $O/pmc_$c.log 2>&1)
  f=$(ls $O/pmc_$c/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" "$c" > $O/pmc_$c.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'lzf' in r['Kernel_Name']:
        k = r['Kernel_Name'][:80]; agg[k] += float(r['Counter_Value']); cnt[k] += 1
for k in agg: print(sys.argv[2], k, 'dispatches', cnt[k], 'sum', agg[k], 'per_dispatch', agg[k] / cnt[k])
PY
done
cat $O/bench_default.json | cut -c1-600; cat $O/kernel_stats.txt | head -8; cat $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt
