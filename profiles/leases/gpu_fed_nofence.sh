#!/bin/bash
# round 6: what a hand-over costs — the release / the acquire fence left out (timing only: the output may be stale), 64 pieces per job
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for L in liblzfear_hip_analysis.so liblzfear_hip_2bcbfddc51.so liblzfear_hip_f7fbdf6a38.so liblzfear_hip_42c9e45030.so; do
for P in 8 64; do
echo -n "$L pieces $P: "; LZF_LIB_PATH=$PWD/rust-lz-fear_amd/$L LZF_FED_PIECES=$P timeout 600 python tools/pmc_decomp.py 240 3 2>&1 | tail -1
done; done
} > gpurun_out/fed_nofence.log 2>&1
cat gpurun_out/fed_nofence.log
