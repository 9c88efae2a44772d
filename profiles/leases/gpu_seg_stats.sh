#!/bin/bash
# per-kernel times of the segmented pipeline (rocprofv3 --kernel-trace --stats) at the given numbers of corpus copies
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
export LZF_SEG_MIN_IN=65536
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/segs; (cd $R && LZF_DECOMPRESS_KERNEL=${VARIANT:-seg} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/segs -- python tools/pmc_decomp.py $c 3 > /tmp/segs.log 2>&1)
  echo "== copies $c"; grep "^jobs" /tmp/segs.log | tail -1
  f=$(ls /tmp/segs/*/*kernel_trace.csv | head -1); python - "$f" <<'PY'
import csv,sys,collections
d=collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name']
    if 'lzf' in n and 'compress_' not in n: d.setdefault(n,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
tot=0
for n,v in d.items():
    v2=sorted(v); print(f"  {n[:64]:64s} calls {len(v):>3d}  min {v2[0]:8.4f}  median {v2[len(v2)//2]:8.4f} ms"); tot+=v2[0]
print(f"  sum of minima {tot:.3f} ms")
PY
done
