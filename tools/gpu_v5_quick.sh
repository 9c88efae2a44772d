# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
for v in v5l256 v5l128 v6l256; do echo "== $v"; LZF_DECOMPRESS_KERNEL=$v timeout 300 python tests/variant_check.py 2>&1 | tail -1; done
echo "== stress v5l128"; LZF_DECOMPRESS_KERNEL=v5l128 timeout 600 python tests/stress_parity.py 3 91 2>&1 | tail -1
LZF_V6_SLICE=16384 bash tools/time_variants.sh 240 paired24 v5l128 v5l256 v5l384 v5s512 v6l256
