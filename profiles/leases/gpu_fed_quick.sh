#!/bin/bash
# round 6: parity of every kernel that shares the copy stage, then the fed call at full size (analysis library)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
A=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so
C=${1:-240}
{
for v in fed paired24; do
  echo -n "$v: "; LZF_LIB_PATH=$A LZF_DECOMPRESS_KERNEL=$v LZF_FED_MIN_IN=1 timeout 600 python tests/variant_check.py 2>&1 | tail -1
done
echo -n "fed, 3 pieces: "; LZF_LIB_PATH=$A LZF_DECOMPRESS_KERNEL=fed LZF_FED_MIN_IN=1 LZF_FED_PIECES=3 timeout 600 python tests/variant_check.py 2>&1 | tail -1
for P in 1 8 16 32 64; do
echo "== $C copies, fed, $P pieces"
LZF_FED_VERBOSE=1 LZF_LIB_PATH=$A LZF_FED_PIECES=$P LZF_VERIFY=1 timeout 600 python tools/pmc_decomp.py $C 3 2>&1 | tail -4
done
echo "== $C copies, nofed"
LZF_LIB_PATH=$A LZF_DECOMPRESS_KERNEL=nofed timeout 600 python tools/pmc_decomp.py $C 3 2>&1 | tail -2
} > gpurun_out/fed_quick.log 2>&1
cat gpurun_out/fed_quick.log
