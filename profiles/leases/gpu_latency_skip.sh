#!/bin/bash
# per-block latency of the pair kernel with one copy-stage phase disabled at a time (LZF_DBG_SKIP analysis builds; output is wrong by design).
# paired256: its parser (12.5 ms per 4 MiB block) is well ahead of the copier, so the differences are the copier's.
cd "$GRAFT_REPO_ROOT" || exit 1
V=${1:-paired256}
for k in 0 2 4 8 16 6 24 30 1; do
  lib=dbg/liblzf_skip$k.so; [ $k = 0 ] && lib=rust-lz-fear_amd/liblzfear_hip_analysis.so
  echo -n "skip $k $V copies 4: "; LZF_LIB_PATH=$PWD/$lib LZF_DECOMPRESS_KERNEL=$V timeout 300 python tools/pmc_decomp.py 4 3 2>&1 | tail -1
done
