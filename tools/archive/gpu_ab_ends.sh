#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
A=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so; B=$PWD/dbg/liblzf_base.so
for v in paired48 paired24 staged16; do LZF_LIB_PATH=$A LZF_DECOMPRESS_KERNEL=$v timeout 300 python tests/variant_check.py 2>&1 | tail -1; done
LZF_LIB_PATH=$A timeout 600 python tests/stress_parity.py 6 9 2>&1 | tail -1
for c in 4 16 240; do for v in paired48 paired24; do for lib in $B $A; do echo -n "copies $c $v $(basename $lib): "; LZF_LIB_PATH=$lib LZF_DECOMPRESS_KERNEL=$v timeout 300 python tools/pmc_decomp.py $c 3 2>&1 | tail -1; done; done; done
