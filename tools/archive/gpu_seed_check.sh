#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02
df -h /dev/shm | tail -1; free -g | head -2
timeout 1500 python -m pytest tests/test_gpu_frame.py -q -m gpu -x 2>&1 | tail -4
LZF_LIB_PATH=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so LZF_FRAME_TRACE=1 timeout 600 python tools/e2e_trace.py 2>&1 | grep -v "^\[frame\] d" | tail -32
E2E_BS=65536 LZF_LIB_PATH=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so LZF_FRAME_TRACE=1 timeout 600 python tools/e2e_trace.py 2>&1 | grep -v "^\[frame\] d" | tail -32
for d in 240 48 1; do timeout 900 python bench.py --copies 240 --distinct $d --no-cpu --no-e2e > gpurun_out/r02/bench_240x$d.log 2>&1; tail -1 gpurun_out/r02/bench_240x$d.log > gpurun_out/r02/bench_240x$d.json; grep "^\[bench\]" gpurun_out/r02/bench_240x$d.log | head -3; cut -c1-260 gpurun_out/r02/bench_240x$d.json; done
