#!/bin/bash
# round 6: slots per CU of the bitmap-fed kernel against the call's time (16 pieces per job), 240 copies
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
A=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so
{
for S in 16 19 20 21 22 24; do
echo -n "slots/CU $S: "; LZF_LIB_PATH=$A LZF_FED_PIECES=16 LZF_FED_SLOTS=$S timeout 600 python tools/pmc_decomp.py 240 3 2>&1 | tail -1
done
} > gpurun_out/fed_slots.log 2>&1
cat gpurun_out/fed_slots.log
