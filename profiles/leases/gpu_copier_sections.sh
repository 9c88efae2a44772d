#!/bin/bash
# cycles the copier of the pair kernel spends in each section of its batch loop (LZF_DBG_PHASE_SEL builds), 196 blocks (latency regime)
cd "$GRAFT_REPO_ROOT" || exit 1
for V in paired256 paired48; do
  echo "== $V: total kilo-cycles per job"; LZF_PRINT_RESERVED=1 LZF_LIB_PATH=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so LZF_DECOMPRESS_KERNEL=$V timeout 300 python tools/pmc_decomp.py 4 2 2>&1 | tail -2
  for k in 0 1 2 3 4 5; do echo -n "section $k: "; LZF_PRINT_RESERVED=1 LZF_LIB_PATH=$PWD/dbg/liblzf_ph$k.so LZF_DECOMPRESS_KERNEL=$V timeout 300 python tools/pmc_decomp.py 4 2 2>&1 | tail -1; done
done
