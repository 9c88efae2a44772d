# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
for v in v4t24 v4t24w6 v4w64; do echo "== $v"; LZF_DECOMPRESS_KERNEL=$v timeout 600 python tests/variant_check.py 2>&1 | tail -2; done
echo "== stress v4t24"; LZF_DECOMPRESS_KERNEL=v4t24 timeout 900 python tests/stress_parity.py 2 11 2>&1 | tail -2
bash tools/time_variants.sh 240 paired24 v4t24 v4t24w6 v4t24w8 v4w64
bash tools/pmc_libs.sh v4t24 rust-lz-fear_amd/liblzfear_hip.so dbg/lib_s2_1.so dbg/lib_s2_2.so dbg/lib_s2_4.so dbg/lib_s2_8.so dbg/lib_s2_16.so dbg/lib_s2_32.so
