#!/bin/bash
# round 6: the bitmap-fed kernel's throughput against its residency (unused LDS per wavefront), 240 copies, one group
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
A=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so
C=${1:-240}
{
for pad in 0 1280 3328 6912 13568; do
  echo -n "pad $pad ($((163840 / (6912 + pad))) per CU): "; LZF_LIB_PATH=$A LZF_FED_GROUPS=1 LZF_FED_PAD_LDS=$pad LZF_PRINT_RESERVED=1 timeout 300 python tools/pmc_decomp.py $C 3 2>&1 | tail -2 | tr '\n' ' '; echo
done
} > gpurun_out/fed_residency.log 2>&1
cat gpurun_out/fed_residency.log
