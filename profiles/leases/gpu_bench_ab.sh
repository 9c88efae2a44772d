#!/bin/bash
# A/B of product-library builds in bench.py's own setting (distinct inputs per copy): profiles/leases/gpu_bench_ab.sh lib1.so lib2.so ...
mkdir -p gpurun_out; rm -f gpurun_out/bench_ab.log
for L in "$@" "$@"; do
  LZF_LIB_PATH=$L timeout 900 python bench.py --no-cpu --no-e2e --no-config4 --no-verify --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print('$L', 'decompress', l['value'], 'compress', l['compress']['value'], 'kernel_ms', l['compress']['roofline']['kernel_ms'])" >> gpurun_out/bench_ab.log
done
cat gpurun_out/bench_ab.log
