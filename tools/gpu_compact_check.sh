#!/bin/bash
# compress kernel change: parity tests of every compress path, then timing (shared-input tool) at 240 and 5 copies
mkdir -p gpurun_out; rm -f gpurun_out/compact_time.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_frame.py -x -q -m gpu -k "compress or silesia or dispatch or frame or writer" > gpurun_out/compact_tests.log 2>&1
echo "exit $?" >> gpurun_out/compact_tests.log; tail -3 gpurun_out/compact_tests.log
for c in 240 5; do
  echo "== copies $c" >> gpurun_out/compact_time.log
  timeout 600 python tools/time_compress.py $c 3 >> gpurun_out/compact_time.log 2>&1
done
grep -v "amdgpu.ids\|^status" gpurun_out/compact_time.log
