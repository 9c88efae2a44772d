#!/bin/bash
# round-2 frame layer check: frame tests + parity tests + a quick end-to-end timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_frame.py -q -m gpu > gpurun_out/r2_frame_tests.log 2>&1
tail -40 gpurun_out/r2_frame_tests.log
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -m gpu > gpurun_out/r2_parity_tests.log 2>&1
tail -40 gpurun_out/r2_parity_tests.log
timeout 600 python bench.py --copies 20 --distinct 4 --steps 2 --warmup 1 --no-cpu 2>&1 | tail -5 > gpurun_out/r2_frame_bench.log
cat gpurun_out/r2_frame_bench.log
