#!/bin/bash
# a candidate product build (liblzfear_hip_try.so) against the committed one: compress parity tests on the candidate, then the bench A/B
mkdir -p gpurun_out
LZF_LIB_PATH=rust-lz-fear_amd/liblzfear_hip_try.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_frame.py -x -q -m gpu -k "compress or silesia or dispatch or frame or writer" > gpurun_out/try_tests.log 2>&1
echo "exit $?" >> gpurun_out/try_tests.log; tail -2 gpurun_out/try_tests.log
bash tools/gpu_bench_ab.sh rust-lz-fear_amd/liblzfear_hip.so rust-lz-fear_amd/liblzfear_hip_try.so
