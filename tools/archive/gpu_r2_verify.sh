#!/bin/bash
# last check of a commit: the whole -m gpu suite, smoke(), and the default bench without its CPU / host legs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r2_gpu_tests.log 2>&1; tail -4 gpurun_out/r2_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --no-cpu --no-e2e 2>&1 | tail -1 | cut -c1-700
