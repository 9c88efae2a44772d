#!/bin/bash
# compress parity (every compress test + the stress round) and the throughput call of the compact kernel
mkdir -p gpurun_out/r05; L=gpurun_out/r05/compact_ab.log; rm -f $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py -x -q -m gpu -k "compress or stress or survey or cap or writer" > gpurun_out/r05/compact_parity.log 2>&1; echo "parity exit $?" >> $L; tail -3 gpurun_out/r05/compact_parity.log >> $L
for c in 96 240; do echo "== copies $c" >> $L; timeout 600 python tools/time_compress.py $c 2 >> $L 2>&1; done
grep -v "amdgpu.ids" $L
