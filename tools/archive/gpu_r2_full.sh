#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r2_gpu_tests.log 2>&1
tail -6 gpurun_out/r2_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_driver_cmd.log 2>&1 ) 2>&1 | grep real
bash tools/refresh_profiles.sh 2>&1 | tail -40
