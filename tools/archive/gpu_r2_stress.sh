#!/bin/bash
# randomised parity stress on the product library (not part of the pytest suite)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python tests/stress_frames_many.py 40 2026 2>&1 | tail -3
  timeout 900 python tests/stress_frames.py 30 77 2>&1 | tail -3
  timeout 900 python tests/stress_parity.py 12 5 2>&1 | tail -3 ) | tee gpurun_out/r2_stress.log
