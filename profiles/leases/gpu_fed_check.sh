#!/bin/bash
# round 6: the bitmap-fed decompress kernel — parity of the forced variant, then A/B against the pair kernel in one launch of N copies
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
A=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so
C=${1:-120}
{
echo "== variant_check fed (every input, min_in 1)"
LZF_LIB_PATH=$A LZF_DECOMPRESS_KERNEL=fed LZF_FED_MIN_IN=1 timeout 600 python tests/variant_check.py 2>&1 | tail -1
for G in 1 2 3 4; do
echo "== $C copies, fed, groups $G"
LZF_LIB_PATH=$A LZF_FED_GROUPS=$G LZF_VERIFY=1 timeout 600 python tools/pmc_decomp.py $C 4 2>&1 | tail -4
done
echo "== $C copies, analysis nofed"
LZF_LIB_PATH=$A LZF_DECOMPRESS_KERNEL=nofed timeout 600 python tools/pmc_decomp.py $C 3 2>&1 | tail -2
echo "== kernel trace (product, $C copies)"
export TMPDIR=/tmp
rm -rf /tmp/fedprof
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/fedprof -o fed -- python tools/pmc_decomp.py $C 3 > /tmp/fedprof.log 2>&1
DB=$(find /tmp/fedprof -name "*_results.db" | head -1); python profiles/summarize_rocpd.py $DB 2>&1 | head -24
} > gpurun_out/fed_check.log 2>&1
cat gpurun_out/fed_check.log
