# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
# per-section shader cycles of the v6 copy kernel (analysis libs dbg/lib_t<k>.so): bash tools/sec_time.sh VARIANT COPIES
v=$1; c=$2
for k in 0 1 2 3 4 5 6 7; do echo -n "section $k: "; LZF_V6_SLICE=16384 LZF_PRINT_RESERVED=1 LZF_LIB_PATH=$GRAFT_REPO_ROOT/dbg/lib_t$k.so LZF_DECOMPRESS_KERNEL=$v timeout 300 python tools/pmc_decomp.py $c 1 2>&1 | tail -2 | tr '\n' ' '; echo; done
