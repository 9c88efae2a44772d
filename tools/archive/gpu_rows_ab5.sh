#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/rows_time.log gpurun_out/rows_stats.log
A=rust-lz-fear_amd/liblzfear_hip_analysis.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compress_u32 or every_compress_kernel or silesia or dictionary or output_full" > gpurun_out/rows_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/rows_tests.log
for c in 240 5; do
  echo "== rows, copies $c" >> gpurun_out/rows_time.log
  LZF_LIB_PATH=$A timeout 600 python tools/time_compress.py $c 2 >> gpurun_out/rows_time.log 2>&1
done
for c in 1 80; do
  echo "== stats, copies $c" >> gpurun_out/rows_stats.log
  LZF_LIB_PATH=rust-lz-fear_amd/liblzfear_hip_dbgrows.so timeout 600 python tools/rows_stats.py $c >> gpurun_out/rows_stats.log 2>&1
done
tail -3 gpurun_out/rows_tests.log
grep -v amdgpu.ids gpurun_out/rows_time.log | grep -v "^status"; grep -v amdgpu.ids gpurun_out/rows_stats.log | grep -v "^ *[0-9]"
