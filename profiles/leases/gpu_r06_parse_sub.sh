#!/bin/bash
# round 6: the parse stage in two sub-chunks of 128-byte regions (16 waves per CU) against one stage of 256-byte regions (8): parity of the
# pipeline's tests, then the bitmap-fed call and the segmented call timed with every job verified; job order natural / longest-first
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
A=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "seg or fed or bitmap or batch_size or malformed or generation" 2>&1 | tail -3
for O in default natural; do
echo "== 240 copies, fed, order $O"
E=""; [ $O = natural ] && E="LZF_DECOMPRESS_ORDER=natural"
env $E LZF_LIB_PATH=$A LZF_VERIFY=1 timeout 600 python tools/pmc_decomp.py 240 4 2>&1 | tail -5
done
for C in 1 4 20; do echo "== $C copies"; LZF_LIB_PATH=$A LZF_VERIFY=1 timeout 600 python tools/pmc_decomp.py $C 5 2>&1 | tail -4; done
} > gpurun_out/parse_sub.log 2>&1
cat gpurun_out/parse_sub.log
