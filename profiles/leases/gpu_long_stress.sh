#!/bin/bash
# long randomised parity runs through the product library (several seeds), beyond the 60 s of the GPU suite
mkdir -p gpurun_out; rm -f gpurun_out/r05_long_stress.log
for seed in 11 12 13; do
  timeout 900 python tests/stress_parity.py 60 $seed 2>&1 | tail -2 >> gpurun_out/r05_long_stress.log
done
for seed in 21 22; do
  timeout 900 python tests/stress_frames.py 30 $seed 2>&1 | tail -2 >> gpurun_out/r05_long_stress.log
done
cat gpurun_out/r05_long_stress.log
