# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
# SQ instruction counters of decompress kernel variants (selected by name, same library):
#   bash tools/pmc_variants2.sh variant...     -> gpurun_out/pmc4/<variant>.txt   (40 copies = 1960 blocks, 468.5 M sequences... see NSEQ)
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc4; cd /tmp; export TMPDIR=/tmp
NSEQ=$((11711759*40))
for v in "$@"; do
  rm -rf $R/gpurun_out/pmc4/$v
  (cd $R && LZF_DECOMPRESS_KERNEL=$v timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc4/$v -- python tools/pmc_decomp.py 40 1 > $R/gpurun_out/pmc4/$v.log 2>&1)
  python $R/tools/pmc_sum.py $R/gpurun_out/pmc4/$v decompress $NSEQ > $R/gpurun_out/pmc4/$v.txt 2>&1
  rm -rf $R/gpurun_out/pmc4/$v
  echo "== $v"; cat $R/gpurun_out/pmc4/$v.txt
done
