# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
# usage (GPU box): bash tools/gpu_v4_check.sh  — parity of the v4 variants + timing vs paired24
for v in v4t24 v4t48 v4t24w8 v4w64 v4w96; do
  echo "== $v" ; LZF_DECOMPRESS_KERNEL=$v timeout 600 python tests/variant_check.py 2>&1 | tail -3
done
echo "== stress v4t24"; LZF_DECOMPRESS_KERNEL=v4t24 timeout 900 python tests/stress_parity.py 3 7 2>&1 | tail -4
echo "== stress v4w64"; LZF_DECOMPRESS_KERNEL=v4w64 timeout 900 python tests/stress_parity.py 3 8 2>&1 | tail -4
bash tools/time_variants.sh 240 paired24 v4t24 v4t48 v4t24w6 v4t24w8 v4w64 v4w96
