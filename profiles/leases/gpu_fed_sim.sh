cd "$GRAFT_REPO_ROOT"; A=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so
LZF_LIB_PATH=$A LZF_FED_GROUPS=1 LZF_SIM_SLOTS=5888 LZF_PRINT_RESERVED=1 timeout 300 python tools/pmc_decomp.py 240 2 2>&1 | tail -3
LZF_LIB_PATH=$A LZF_FED_GROUPS=1 LZF_FED_PAD_LDS=3328 LZF_SIM_SLOTS=4096 LZF_PRINT_RESERVED=1 timeout 300 python tools/pmc_decomp.py 240 2 2>&1 | tail -3
