#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_frame.py -q -m gpu 2>&1 | tail -15
LZF_LIB_PATH=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so LZF_FRAME_TRACE=1 timeout 600 python tools/e2e_trace.py > gpurun_out/r2_e2e_trace.log 2>&1
grep -v "^\[frame\] c" gpurun_out/r2_e2e_trace.log | tail -30
