# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
# usage: bash tools/time_libs.sh VARIANT COPIES lib.so...   — kernel time of one variant across analysis libraries
v=$1; c=$2; shift; shift
for lib in "$@"; do echo -n "$(basename $lib .so): "; LZF_LIB_PATH=$GRAFT_REPO_ROOT/$lib LZF_DECOMPRESS_KERNEL=$v timeout 300 python tools/pmc_decomp.py $c 2 2>&1 | tail -1; done
