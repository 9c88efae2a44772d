# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
# usage (GPU box): bash tools/gpu_walk_check.sh  — parity of the walk variants + timing vs paired24
mkdir -p gpurun_out
for v in walk48 walk64 walk96 walk128; do
  echo "== $v" ; LZF_DECOMPRESS_KERNEL=$v timeout 600 python tests/variant_check.py 2>&1 | tail -3
done
bash tools/time_variants.sh 240 paired24 walk48 walk64 walk96 walk128
