#!/bin/bash
export LZF_LIB_PATH="${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so"
cd "$GRAFT_REPO_ROOT" || exit 1
for c in 4 16; do echo "copies $c:"; for v in paired48 walk48 walk64 walk96 walk128 v4t48 v4w64 v4w96 v5s512 staged16 staged32 direct4w; do echo -n "$v: "; LZF_DECOMPRESS_KERNEL=$v timeout 300 python tools/pmc_decomp.py $c 3 2>&1 | tail -1; done; done
