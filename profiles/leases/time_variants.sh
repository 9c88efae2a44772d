# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
# usage: bash profiles/leases/time_variants.sh COPIES variant...
c=$1; shift
for v in "$@"; do echo -n "$v: "; LZF_DECOMPRESS_KERNEL=$v timeout 300 python tools/pmc_decomp.py $c 2 2>&1 | tail -1; done
