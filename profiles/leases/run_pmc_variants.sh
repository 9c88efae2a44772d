# usage: bash profiles/leases/run_pmc_variants.sh variant...   (dbg/lib_<variant>.so built with -DLZF_DBG_SKIP=...)
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc2; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  (cd $R && LZF_LIB_PATH=$R/dbg/lib_$v.so timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc2/$v -- python tools/pmc_decomp.py 40 1 > $R/gpurun_out/pmc2/$v.log 2>&1)
  (cd $R && LZF_LIB_PATH=$R/dbg/lib_$v.so timeout 300 python tools/pmc_decomp.py 240 2 > $R/gpurun_out/pmc2/${v}_time.log 2>&1)
  echo -n "$v: "; tail -1 $R/gpurun_out/pmc2/${v}_time.log
done
