#!/bin/bash
# round 5: is the compact compress kernel issue-bound or latency-bound at full residency?  SQ wait / active counters of one launch
# over 96 copies (4 896 blocks > the 4 608 the chip holds), and the instruction counters of the same launch.  Through gpurun.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  (cd $R && timeout 500 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcw_$tag -- python tools/time_compress.py 96 1 > $O/pmcw_$tag.log 2>&1)
  f=$(ls $O/pmcw_$tag/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if 'compress_compact_kernel<false>' in r['Kernel_Name'] or 'compress_compact_kernel<(bool)0>' in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(agg.items()): print(f"{k:28s} {v:.4g}")
PY
  rm -rf $O/pmcw_$tag
done 2>&1 | tee $O/pmc_compress_wait.txt
grep "jobs" $O/pmcw_*.log | head -3
