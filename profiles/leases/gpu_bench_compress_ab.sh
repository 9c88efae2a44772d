#!/bin/bash
# A/B of compress-kernel builds in bench.py's own setting (240 copies with distinct inputs): profiles/leases/gpu_bench_compress_ab.sh <lib.so> ...
mkdir -p gpurun_out/r05; L=gpurun_out/r05/bench_compress_ab.log; rm -f $L
for lib in "$@"; do
  echo "== $lib" >> $L
  LZF_LIB_PATH=$lib timeout 900 python bench.py --no-cpu --no-e2e --no-config4 --no-config5 --steps 2 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
c = d['compress']
print('compress', c['value'], 'GiB/s', c['ms_per_step'], 'ms; kernel', c['roofline']['kernel_ms'], '| decompress', d['value'])" >> $L 2>&1
done
cat $L
