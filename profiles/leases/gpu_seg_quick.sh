# quick look at the segmented pipeline after a kernel change: timing at 1 / 4 / 15 copies (every job verified), the resolver's timers, parity
export LZF_SEG_MIN_IN=65536
L=${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so
for c in ${COPIES:-1 4 15}; do echo -n "copies $c: "; LZF_LIB_PATH=$L LZF_PRINT_RESERVED=1 LZF_VERIFY=1 LZF_DECOMPRESS_KERNEL=seg timeout 300 python tools/pmc_decomp.py $c 3 2>&1 | tail -3 | tr "\n" " "; echo; done
[ -f ab/liblzfear_segtime.so ] && LZF_LIB_PATH=$PWD/ab/liblzfear_segtime.so timeout 300 python tools/seg_debug.py --big 8 2>&1 | grep "ok (\|FAIL" | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "seg or variant or mixed" 2>&1 | tail -3
