#!/bin/bash
# round 5: BASELINE configs[4] — the bench leg alone and its rocprofv3 kernel statistics
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R && timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 > $O/bench_config5.log 2>&1; tail -1 $O/bench_config5.log > $O/bench_config5.json
cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof5 -o x -- python bench.py --workload config5 --steps 2 --warmup 1 --no-verify > $O/prof5.log 2>&1)
DB=$(ls $O/prof5/*/x_results.db $O/prof5/x_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $O/config5_kernel_stats.txt
rm -rf $O/prof5
tail -3 $O/bench_config5.log | cut -c1-3000; head -30 $O/config5_kernel_stats.txt
