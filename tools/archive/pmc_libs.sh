# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
# SQ instruction counters of one decompress variant across analysis libraries:
#   bash tools/pmc_libs.sh VARIANT lib.so...     -> gpurun_out/pmc5/<libname>.txt
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc5; cd /tmp; export TMPDIR=/tmp
NSEQ=$((11711759*40))
v=$1; shift
for lib in "$@"; do
  n=$(basename $lib .so)
  rm -rf $R/gpurun_out/pmc5/$n
  (cd $R && LZF_LIB_PATH=$R/$lib LZF_DECOMPRESS_KERNEL=$v timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc5/$n -- python tools/pmc_decomp.py 40 1 > $R/gpurun_out/pmc5/$n.log 2>&1)
  python $R/tools/pmc_sum.py $R/gpurun_out/pmc5/$n decompress $NSEQ > $R/gpurun_out/pmc5/$n.txt 2>&1
  rm -rf $R/gpurun_out/pmc5/$n
  echo "== $n"; grep -E "INSTS_(VALU|SALU|LDS)|WAVE_CYCLES" $R/gpurun_out/pmc5/$n.txt
done
