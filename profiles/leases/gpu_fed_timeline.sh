#!/bin/bash
# round 6: jobs of the bitmap-fed kernel running over time (-DLZF_DBG_TIMELINE build), 240 copies
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
LZF_FED_VERBOSE=1 LZF_LIB_PATH=$PWD/rust-lz-fear_amd/liblzfear_hip_5c5d1016b1.so LZF_TIMELINE=1 timeout 300 python tools/pmc_decomp.py ${1:-240} 2 > gpurun_out/fed_timeline.log 2>&1
cat gpurun_out/fed_timeline.log
