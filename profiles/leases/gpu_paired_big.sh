#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
A=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so
for v in paired128 paired256; do LZF_LIB_PATH=$A LZF_DECOMPRESS_KERNEL=$v timeout 300 python tests/variant_check.py 2>&1 | tail -1; done
for v in paired48 paired128 paired256; do
  for lib in $A $PWD/dbg/liblzf_skip1.so; do echo -n "$v $(basename $lib): "; LZF_LIB_PATH=$lib LZF_DECOMPRESS_KERNEL=$v timeout 300 python tools/pmc_decomp.py 4 3 2>&1 | tail -1; done
done
