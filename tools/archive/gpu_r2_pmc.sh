#!/bin/bash
# only the HBM-traffic passes of tools/refresh_profiles.sh (the rest of gpurun_out/r02 is kept)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $R && timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --copies 48 --distinct 1 --steps 1 --warmup 0 --no-cpu --no-e2e --no-verify > $O/pmc_$c.log 2>&1)
  f=$(ls $O/pmc_$c/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" "$c" > $O/pmc_$c.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter(); grid = {}
for r in csv.DictReader(open(sys.argv[1])):
    if 'lzf' in r['Kernel_Name']:
        k = r['Kernel_Name'][:90]; agg[k] += float(r['Counter_Value']); cnt[k] += 1; grid[k] = r.get('Grid_Size', '')
for k in agg: print(sys.argv[2], '|', k, '| dispatches', cnt[k], '| grid', grid[k], '| sum', agg[k], '| per_dispatch', agg[k] / cnt[k])
PY
  rm -rf $O/pmc_$c
done
cat $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt; tail -2 $O/pmc_FETCH_SIZE.log | cut -c1-1500
