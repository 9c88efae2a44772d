#!/bin/bash
# round 5: the team compress kernel (latency class) on a GPU box — parity through the compress tests, then the 51- and 255-block
# calls against the compact kernel (LZF_COMPRESS_TEAM_MAX=0 in the analysis flavour turns the class off).
mkdir -p gpurun_out; L=gpurun_out/team_check.log; rm -f $L
A=rust-lz-fear_amd/liblzfear_hip_analysis.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compress" > gpurun_out/team_parity.log 2>&1; echo "parity exit $?" >> $L; tail -3 gpurun_out/team_parity.log >> $L
for c in 1 5; do
  echo "== team, copies $c" >> $L
  LZF_LIB_PATH=$A timeout 300 python tools/time_compress.py $c 3 >> $L 2>&1
  echo "== compact, copies $c" >> $L
  LZF_LIB_PATH=$A LZF_COMPRESS_TEAM_MAX=0 timeout 300 python tools/time_compress.py $c 3 >> $L 2>&1
done
grep -v "amdgpu.ids" $L
D=$(ls rust-lz-fear_amd/liblzfear_hip_c6c2f81bec.so 2>/dev/null)
if [ -n "$D" ]; then LZF_LIB_PATH=$D timeout 300 python tools/team_stats.py > gpurun_out/team_stats.log 2>&1; grep -v amdgpu.ids gpurun_out/team_stats.log | head -12; fi
