#!/bin/bash
# round 6: pieces per job against the call's time, 240 copies (hand-overs inside one XCD)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
A=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so
C=${1:-240}
{
echo -n "fed3: "; LZF_FED_VERBOSE=1 LZF_LIB_PATH=$A LZF_DECOMPRESS_KERNEL=fed LZF_FED_MIN_IN=1 LZF_FED_PIECES=3 timeout 600 python tests/variant_check.py 2>&1 | tail -2
for P in 1 8 16 32 64 128; do
echo -n "== $C copies, fed, $P pieces: "
LZF_LIB_PATH=$A LZF_FED_PIECES=$P LZF_VERIFY=1 timeout 600 python tools/pmc_decomp.py $C 3 2>&1 | tail -2 | tr '\n' ' '; echo
done
} > gpurun_out/fed_wt.log 2>&1
cat gpurun_out/fed_wt.log
