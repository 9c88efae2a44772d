#!/bin/bash
# round 6: per-kernel durations of one decompress call (tools/pmc_decomp.py COPIES) under rocprofv3 --kernel-trace; LZF_LIB_PATH selects the library
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export TMPDIR=/tmp; C=${1:-240}
mkdir -p gpurun_out
rm -rf /tmp/ks; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -- python tools/pmc_decomp.py $C 3 > /tmp/ks.log 2>&1
grep "^jobs" /tmp/ks.log | tail -2
python - <<'PY'
import csv, glob, collections
kt = glob.glob('/tmp/ks/*/*kernel_trace.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    n = r['Kernel_Name']
    if 'lzf' in n: d[n.split('(')[0][-60:]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k:62s} calls {len(v):3d}  min {v2[0]:8.3f}  median {v2[len(v2)//2]:8.3f}  max {v2[-1]:8.3f} ms")
PY
