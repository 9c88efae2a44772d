#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
LZF_FED_SLOTS=0 LZF_LIB_PATH=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so LZF_PER_BLOCK=1 timeout 600 python tools/pmc_decomp.py ${1:-100} 2 > gpurun_out/fed_perblock.log 2>&1
cat gpurun_out/fed_perblock.log
