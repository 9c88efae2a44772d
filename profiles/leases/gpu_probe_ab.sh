#!/bin/bash
# launch order of the compress batch: the cost probe's sample size / count (analysis knob LZF_PROBE) and no order at all
mkdir -p gpurun_out; rm -f gpurun_out/probe_ab.log
A=rust-lz-fear_amd/liblzfear_hip_analysis.so
for p in 65536,1 16384,4 32768,4 65536,4 131072,1 262144,1 32768,2; do
  echo "== LZF_PROBE=$p" >> gpurun_out/probe_ab.log
  LZF_LIB_PATH=$A LZF_PROBE=$p timeout 600 python tools/compress_utilisation.py 240 2>&1 | grep "2.4 GHz and 18" >> gpurun_out/probe_ab.log
done
echo "== natural order" >> gpurun_out/probe_ab.log
LZF_LIB_PATH=$A LZF_COMPRESS_ORDER=natural timeout 600 python tools/compress_utilisation.py 240 2>&1 | grep "2.4 GHz and 18" >> gpurun_out/probe_ab.log
cat gpurun_out/probe_ab.log
