# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
for bs in 65536 262144 1048576; do for c in 16; do echo "block $bs copies $c:"; for v in auto v6l256 v6l128; do echo -n "$v: "; LZF_BS=$bs LZF_DECOMPRESS_KERNEL=$v timeout 300 python tools/pmc_decomp.py $c 3 2>&1 | tail -1; done; done; done
