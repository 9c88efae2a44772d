#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_frame.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 900 python tests/stress_frames_many.py 40 777 2>&1 | tail -2
timeout 900 python tests/stress_frames.py 20 99 2>&1 | tail -1
LZF_LIB_PATH=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so timeout 600 python tools/e2e_trace.py 2>&1 | grep "_many call" | tail -6
