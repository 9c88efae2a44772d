#!/bin/bash
# phase traces of the frame drivers from host buffers (analysis library, LZF_FRAME_TRACE) at both block sizes + the config4 line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02
LZF_LIB_PATH=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so LZF_FRAME_TRACE=1 timeout 600 python tools/e2e_trace.py > gpurun_out/r02/frame_e2e_trace_4MiB.txt 2>&1
E2E_BS=65536 LZF_LIB_PATH=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so LZF_FRAME_TRACE=1 timeout 600 python tools/e2e_trace.py > gpurun_out/r02/frame_e2e_trace_64KiB.txt 2>&1
grep "_many call" gpurun_out/r02/frame_e2e_trace_4MiB.txt gpurun_out/r02/frame_e2e_trace_64KiB.txt
timeout 900 python bench.py --workload config4 > gpurun_out/r02/bench_config4.log 2>&1; tail -1 gpurun_out/r02/bench_config4.log > gpurun_out/r02/bench_config4.json; cut -c1-200 gpurun_out/r02/bench_config4.json
