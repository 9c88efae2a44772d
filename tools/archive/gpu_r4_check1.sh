#!/bin/bash
# round 4, first check: hardening tests of the decompress path, the compress variants, and the rows / compact A/B on one box
mkdir -p gpurun_out; rm -f gpurun_out/r4_*.log
timeout 1500 python -m pytest tests/test_gpu_hardening.py -x -q -m gpu -s > gpurun_out/r4_hardening.log 2>&1
echo "exit $?" >> gpurun_out/r4_hardening.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_compress_kernel or segmented" > gpurun_out/r4_parity_subset.log 2>&1
echo "exit $?" >> gpurun_out/r4_parity_subset.log
A=rust-lz-fear_amd/liblzfear_hip_analysis.so
for k in compact rows; do for c in 240 5; do
  echo "== $k, copies $c" >> gpurun_out/r4_rows_time.log
  LZF_LIB_PATH=$A LZF_COMPRESS_KERNEL=$k timeout 600 python tools/time_compress.py $c 2 >> gpurun_out/r4_rows_time.log 2>&1
done; done
tail -4 gpurun_out/r4_hardening.log; tail -3 gpurun_out/r4_parity_subset.log; grep -v "amdgpu.ids\|^status" gpurun_out/r4_rows_time.log
