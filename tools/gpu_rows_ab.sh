#!/bin/bash
# row-mapped compress kernel against the one-block-per-wave compact kernel: parity tests, then timings of both on the same box
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compress or dispatch or silesia" > gpurun_out/rows_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/rows_tests.log
A=rust-lz-fear_amd/liblzfear_hip_analysis.so
for c in 240 5; do
  echo "== rows, copies $c" >> gpurun_out/rows_time.log
  LZF_LIB_PATH=$A timeout 600 python tools/time_compress.py $c 3 >> gpurun_out/rows_time.log 2>&1
  echo "== compact, copies $c" >> gpurun_out/rows_time.log
  LZF_LIB_PATH=$A LZF_COMPRESS_KERNEL=compact timeout 600 python tools/time_compress.py $c 3 >> gpurun_out/rows_time.log 2>&1
done
tail -5 gpurun_out/rows_tests.log; cat gpurun_out/rows_time.log
