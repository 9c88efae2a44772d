#!/bin/bash
# the segmented pipeline in groups (capi.hip: seg_groups): call time by grouping (LZF_SEG_GROUPS, analysis library; every job verified), then parity
export LZF_SEG_MIN_IN=65536
L=${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so
mkdir -p gpurun_out/r05; O=gpurun_out/r05/seg_groups.log; rm -f $O
for c in ${COPIES:-4 10 20}; do
  for g in ${GROUPS_LIST:-100 50,50 40,60 30,70 25,35,40 33,33,34 20,30,50 15,25,60 25,25,25,25 10,20,30,40}; do
    echo -n "copies $c groups $g: " >> $O
    LZF_SEG_GROUPS=$g LZF_LIB_PATH=$L LZF_VERIFY=1 LZF_DECOMPRESS_KERNEL=seg timeout 300 python tools/pmc_decomp.py $c 6 > /tmp/sg.log 2>&1; (grep "^jobs" /tmp/sg.log | sort -k6 -n | head -1 | tr "\n" " "; grep "^verify" /tmp/sg.log) >> $O
  done
done
cat $O
[ -n "${SKIP_TESTS:-}" ] || timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_frame.py -m gpu -x -q -k "seg or variant or mixed or many or sized or dispatch" 2>&1 | tail -3
