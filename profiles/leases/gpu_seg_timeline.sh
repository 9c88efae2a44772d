#!/bin/bash
# start / end of every kernel of the last decompress call of tools/pmc_decomp.py <copies> (rocprofv3 --kernel-trace), ms from the first
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
export LZF_SEG_MIN_IN=65536
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp
rm -rf /tmp/segs; (cd $R && LZF_DECOMPRESS_KERNEL=${VARIANT:-seg} timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/segs -- python tools/pmc_decomp.py ${1:-4} 3 > /tmp/segs.log 2>&1)
grep "^jobs" /tmp/segs.log | tail -1
f=$(ls /tmp/segs/*/*kernel_trace.csv | head -1); python - "$f" <<'PY'
import csv,sys
ev=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name']
    if 'lzf' in n and 'compress_' not in n: ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),n.split('(')[0][-40:], r.get('Queue_Id','')))
ev.sort()
# last call: after the last gap > 2 ms
cut=0
for i in range(1,len(ev)):
    if ev[i][0]-max(e[1] for e in ev[:i])>2_000_000: cut=i
ev=ev[cut:]; t0=ev[0][0]
for s,e,n,q in ev: print(f"{(s-t0)/1e6:8.3f} .. {(e-t0)/1e6:8.3f}  q{q}  {n}")
PY
