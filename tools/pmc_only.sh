set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  (cd $R && LZF_DECOMPRESS_KERNEL=paired24 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --copies 40 --steps 1 --warmup 0 --no-cpu --no-verify > $O/pmc_$c.log 2>&1)
  f=$(ls $O/pmc_$c/*/*_counter_collection.csv 2>/dev/null | head -1)
  python - "$f" "$c" > $O/pmc_$c.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'lzf' in r['Kernel_Name']:
        k = r['Kernel_Name'][:80]; agg[k] += float(r['Counter_Value']); cnt[k] += 1
for k in agg: print(sys.argv[2], k, 'dispatches', cnt[k], 'sum', agg[k], 'per_dispatch', agg[k] / cnt[k])
PY
done
cat $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt | grep paired
