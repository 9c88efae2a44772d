#!/bin/bash
# candidate build against the committed one for the decompress side: parity subset on the candidate, bench A/B, pair-kernel variants at full size
mkdir -p gpurun_out
LZF_LIB_PATH=rust-lz-fear_amd/liblzfear_hip_try.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decompress or dispatch or silesia or segmented or hc_fixtures or oversized" > gpurun_out/try_tests.log 2>&1
echo "exit $?" >> gpurun_out/try_tests.log; tail -2 gpurun_out/try_tests.log
bash tools/gpu_bench_ab.sh rust-lz-fear_amd/liblzfear_hip.so rust-lz-fear_amd/liblzfear_hip_try.so
for v in paired24 paired16 staged16; do echo -n "try $v: "; LZF_LIB_PATH=rust-lz-fear_amd/liblzfear_hip_try_analysis.so LZF_DECOMPRESS_KERNEL=$v timeout 300 python tools/pmc_decomp.py 240 2 2>&1 | tail -1; done
for v in paired24 paired16 staged16; do echo -n "base $v: "; LZF_LIB_PATH=rust-lz-fear_amd/liblzfear_hip_analysis.so LZF_DECOMPRESS_KERNEL=$v timeout 300 python tools/pmc_decomp.py 240 2 2>&1 | tail -1; done
