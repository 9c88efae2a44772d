#!/bin/bash
# compact compress kernel in bench.py's setting with fewer resident wavefronts per CU (analysis library: LZF_COMPACT_PAD_LDS = unused LDS per wave)
mkdir -p gpurun_out/r05; L=gpurun_out/r05/compact_residency.log; rm -f $L
LIB=${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so
for pad in ${PADS:-0 1024 2048 3072 5120 8192}; do
  echo -n "pad $pad: " >> $L
  LZF_COMPACT_PAD_LDS=$pad LZF_LIB_PATH=$LIB timeout 900 python bench.py --no-cpu --no-e2e --no-config4 --no-config5 --steps 2 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
c = d['compress']
print('compress', c['value'], 'GiB/s', c['ms_per_step'], 'ms; kernel', c['roofline']['kernel_ms'], '| decompress', d['value'])" >> $L 2>&1
done
cat $L
