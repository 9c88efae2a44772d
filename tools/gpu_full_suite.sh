#!/bin/bash
# the driver's round-end sequence on one box: pytest -m gpu, smoke()
mkdir -p gpurun_out/suite
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/suite/gpu_suite.log 2>&1
echo "exit $?" >> gpurun_out/suite/gpu_suite.log
tail -60 gpurun_out/suite/gpu_suite.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/suite/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/suite/smoke.log
