#!/bin/bash
# round 6: the team compress kernel before / after the carried-table work, 255-block call (5 copies of the corpus), same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for L in liblzfear_hip_oldteam_dbg.so liblzfear_hip_analysis.so liblzfear_hip_oldteam_dbg.so liblzfear_hip_analysis.so; do
echo -n "$L: "; LZF_LIB_PATH=$PWD/rust-lz-fear_amd/$L timeout 300 python tools/time_compress.py 5 4 2>&1 | grep "^jobs" | tail -2 | tr '\n' ' '; echo
done
} > gpurun_out/team_ab.log 2>&1
cat gpurun_out/team_ab.log
