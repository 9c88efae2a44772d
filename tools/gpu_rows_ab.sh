#!/bin/bash
# A/B of row-mapped compress kernel variants against the compact kernel on one box: tools/gpu_rows_ab.sh <lib.so> ...
# (every lib is an analysis flavour; LZF_COMPRESS_KERNEL=rows selects the variant under test)
mkdir -p gpurun_out; rm -f gpurun_out/rows_time.log
A=rust-lz-fear_amd/liblzfear_hip_analysis.so
echo "== compact" >> gpurun_out/rows_time.log
LZF_LIB_PATH=$A LZF_COMPRESS_KERNEL=compact timeout 600 python tools/time_compress.py 240 2 >> gpurun_out/rows_time.log 2>&1
for L in "$@"; do for c in 240 5; do
  echo "== rows $L, copies $c" >> gpurun_out/rows_time.log
  LZF_LIB_PATH=$L LZF_COMPRESS_KERNEL=rows timeout 600 python tools/time_compress.py $c 2 >> gpurun_out/rows_time.log 2>&1
done; done
grep -v "amdgpu.ids\|^status" gpurun_out/rows_time.log
