# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
# LDS counters of a decompress variant: bash tools/pmc_lds.sh VARIANT
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc7; cd /tmp; export TMPDIR=/tmp
NSEQ=$((11711759*40))
v=$1
rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_INSTS_LDS[A-Z_]*\|SQ_ACTIVE_INST_LDS\|SQ_WAIT_INST_LDS\|SQ_INST_CYCLES[A-Z_]*" | sort -u | tr '\n' ' '; echo
i=0
for grp in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc7/g$i
  (cd $R && LZF_V6_SLICE=16384 LZF_DECOMPRESS_KERNEL=$v timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc7/g$i -- python tools/pmc_decomp.py 40 1 > $R/gpurun_out/pmc7/g$i.log 2>&1)
  python $R/tools/pmc_sum.py $R/gpurun_out/pmc7/g$i "" $NSEQ | grep -A20 "decompress\|v6_copy\|v6_parse" | grep -v "^--"
  tail -2 $R/gpurun_out/pmc7/g$i.log | head -1
  rm -rf $R/gpurun_out/pmc7/g$i
done
