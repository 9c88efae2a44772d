#!/bin/bash
# per-block latency of the pair kernel with one copy-stage phase disabled at a time (LZF_DBG_SKIP analysis builds; output is wrong by design)
cd "$GRAFT_REPO_ROOT" || exit 1
for k in 0 2 4 8 16 30 1; do
  lib=dbg/liblzf_skip$k.so; [ $k = 0 ] && lib=rust-lz-fear_amd/liblzfear_hip_analysis.so
  for v in paired48; do echo -n "skip $k $v copies 4: "; LZF_LIB_PATH=$PWD/$lib LZF_DECOMPRESS_KERNEL=$v timeout 300 python tools/pmc_decomp.py 4 3 2>&1 | tail -1; done
done
