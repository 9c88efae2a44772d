#!/bin/bash
# round 6: cycles a wave of the bitmap-fed kernel spends in each section (tools/build_phase_variants.py), full occupancy
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
C=${1:-240}
L=(liblzfear_hip_353b749d26.so liblzfear_hip_ac4e409dc6.so liblzfear_hip_1bb935d68d.so liblzfear_hip_99860d22ba.so liblzfear_hip_a9cc1c2ecf.so liblzfear_hip_f1c2aa95a2.so liblzfear_hip_64b60df025.so liblzfear_hip_a106876f47.so liblzfear_hip_1e10ecf9d0.so)
N=("between batches" "set-up + checks" "far issue + literals" "far stores" "match rounds" "flush" "stage + bit map" "bit map -> list" "lengths + chain check")
{
echo -n "total: "; LZF_LIB_PATH=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so LZF_FED_GROUPS=1 LZF_PRINT_RESERVED=1 timeout 300 python tools/pmc_decomp.py $C 2 2>&1 | tail -2 | tr '\n' ' '; echo
for k in 0 1 2 3 4 5 6 7 8; do
  echo -n "section $k (${N[$k]}): "; LZF_LIB_PATH=$PWD/rust-lz-fear_amd/${L[$k]} LZF_FED_GROUPS=1 LZF_PRINT_RESERVED=1 timeout 300 python tools/pmc_decomp.py $C 2 2>&1 | tail -1
done
} > gpurun_out/fed_sections.log 2>&1
cat gpurun_out/fed_sections.log
