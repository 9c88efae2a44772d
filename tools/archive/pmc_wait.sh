# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
# SQ wait/active counters of a decompress variant: bash tools/pmc_wait.sh VARIANT [lib]
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc6; cd /tmp; export TMPDIR=/tmp
NSEQ=$((11711759*40))
v=$1; lib=${2:-rust-lz-fear_amd/liblzfear_hip.so}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc6/g$i
  (cd $R && LZF_LIB_PATH=$R/$lib LZF_DECOMPRESS_KERNEL=$v timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc6/g$i -- python tools/pmc_decomp.py 40 1 > $R/gpurun_out/pmc6/g$i.log 2>&1)
  python $R/tools/pmc_sum.py $R/gpurun_out/pmc6/g$i decompress $NSEQ | grep -v "^void"
  rm -rf $R/gpurun_out/pmc6/g$i
done
