# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
for v in v6l256; do echo "== $v"; LZF_V6_PARTS=2 LZF_DECOMPRESS_KERNEL=$v timeout 300 python tests/variant_check.py 2>&1 | tail -1; done
for k in 1 2 3 4 6; do echo -n "parts $k: "; LZF_V6_PARTS=$k bash tools/time_variants.sh 240 v6l256; done
bash tools/time_variants.sh 240 paired24
