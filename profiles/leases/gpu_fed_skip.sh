#!/bin/bash
# round 6: where the bitmap-fed kernel's time goes — one copy-stage phase disabled per build (tools/build_skip_variants.py), 240 copies, one run of the call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
C=${1:-240}
declare -A L=( [0]=liblzfear_hip_analysis.so [1]=liblzfear_hip_6e81ec8169.so [2]=liblzfear_hip_8880e47d41.so [4]=liblzfear_hip_2cfba0be85.so [8]=liblzfear_hip_682238eecf.so [16]=liblzfear_hip_f0c9bfbe61.so [30]=liblzfear_hip_cdbe932c22.so )
{
for k in 0 1 30 2 4 8 16; do
  echo -n "skip $k: "; LZF_LIB_PATH=$PWD/rust-lz-fear_amd/${L[$k]} LZF_FED_GROUPS=1 LZF_PRINT_RESERVED=1 timeout 300 python tools/pmc_decomp.py $C 3 2>&1 | tail -2 | tr '\n' ' '; echo
done
} > gpurun_out/fed_skip.log 2>&1
cat gpurun_out/fed_skip.log
