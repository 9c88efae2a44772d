# segmented pipeline vs pair kernel at small batch sizes (analysis library: LZF_DECOMPRESS_KERNEL=seg | noseg), then a kernel trace
export LZF_LIB_PATH="${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so"
export LZF_SEG_MIN_IN=65536
for c in 1 4 20; do
  for v in seg noseg; do echo -n "copies $c $v: "; LZF_VERIFY=1 LZF_DECOMPRESS_KERNEL=$v timeout 300 python tools/pmc_decomp.py $c 3 2>&1 | tail -2 | tr '\n' ' '; echo; done
done
cd /tmp && export TMPDIR=/tmp
LZF_DECOMPRESS_KERNEL=seg timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_seg -o seg -- python $GRAFT_REPO_ROOT/tools/pmc_decomp.py 4 3 > /dev/null 2>&1
DB=$(find /tmp/prof_seg -name "*results.db" | head -1); python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py $DB 2>&1 | head -40
