#!/bin/bash
# one-wave kernels (parse then copy, serial per chunk): full vs parse-only (LZF_DBG_SKIP=1) -> latency of the copy stage alone
cd "$GRAFT_REPO_ROOT" || exit 1
for v in staged32 staged16 direct4w; do
  for lib in rust-lz-fear_amd/liblzfear_hip_analysis.so dbg/liblzf_skip1.so; do
    echo -n "$v $(basename $lib): "; LZF_LIB_PATH=$PWD/$lib LZF_DECOMPRESS_KERNEL=$v timeout 300 python tools/pmc_decomp.py 4 3 2>&1 | tail -1
  done
done
