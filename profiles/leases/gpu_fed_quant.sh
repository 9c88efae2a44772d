cd "$GRAFT_REPO_ROOT"; A=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so
for c in 109 219 240 328; do echo -n "copies $c fed: "; LZF_LIB_PATH=$A LZF_FED_GROUPS=1 timeout 300 python tools/pmc_decomp.py $c 3 2>&1 | tail -1; done
for c in 67 135 203 240; do echo -n "copies $c nofed: "; LZF_LIB_PATH=$A LZF_DECOMPRESS_KERNEL=nofed timeout 300 python tools/pmc_decomp.py $c 3 2>&1 | tail -1; done
