#!/bin/bash
# round 6: the bitmap-fed call as two halves (LZF_FED_HALVES = slots per CU of the first half's copy stage, under which the second half's
# parse runs): every job verified, then bench.py's setting, same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
A=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so
{
for H in 0 16; do
echo "== tool, 240 copies, halves $H"
LZF_FED_HALVES=$H LZF_DECOMPRESS_ORDER=natural LZF_LIB_PATH=$A LZF_VERIFY=1 timeout 600 python tools/pmc_decomp.py 240 4 2>&1 | tail -3
done
} > gpurun_out/halves.log 2>&1
cat gpurun_out/halves.log
bash profiles/leases/gpu_r06_bench_ab.sh "liblzfear_hip_analysis.so" "LZF_FED_HALVES=12 LZF_DECOMPRESS_ORDER=natural liblzfear_hip_analysis.so" "LZF_FED_HALVES=16 LZF_DECOMPRESS_ORDER=natural liblzfear_hip_analysis.so" "LZF_FED_HALVES=18 LZF_DECOMPRESS_ORDER=natural liblzfear_hip_analysis.so" "LZF_FED_HALVES=21 LZF_DECOMPRESS_ORDER=natural liblzfear_hip_analysis.so"
