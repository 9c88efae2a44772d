# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
# SQ instruction counters of the v6 copy kernel across analysis libraries: bash tools/pmc_libs6.sh VARIANT lib.so...
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc8; cd /tmp; export TMPDIR=/tmp; export LZF_V6_SLICE=16384
NSEQ=$((11711759*40))
v=$1; shift
for lib in "$@"; do
  n=$(basename $lib .so); rm -rf $R/gpurun_out/pmc8/$n
  (cd $R && LZF_LIB_PATH=$R/$lib LZF_DECOMPRESS_KERNEL=$v timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL --output-format csv -d $R/gpurun_out/pmc8/$n -- python tools/pmc_decomp.py 40 1 > $R/gpurun_out/pmc8/$n.log 2>&1)
  echo "== $n"; python $R/tools/pmc_sum.py $R/gpurun_out/pmc8/$n v6_copy $NSEQ | grep -E "INSTS_(VALU|SALU|LDS)|LDS_IDX|UNALIGNED"
  rm -rf $R/gpurun_out/pmc8/$n
done
