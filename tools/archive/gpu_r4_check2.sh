#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_hardening.py -x -q -m gpu -k "geometry or fallbacks or damaged" > gpurun_out/r4_hardening2.log 2>&1
echo "exit $?" >> gpurun_out/r4_hardening2.log
tail -5 gpurun_out/r4_hardening2.log
bash tools/gpu_lane_histogram.sh
