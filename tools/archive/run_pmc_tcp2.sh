# TA_BUSY / TD_BUSY / TCP access counters for the LZF_DBG_SKIP analysis builds dbg/lib_<variant>.so: bash tools/run_pmc_tcp2.sh variant...
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc4; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  (cd $R && LZF_LIB_PATH=$R/dbg/lib_$v.so timeout 200 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TCP_TOTAL_ACCESSES_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $R/gpurun_out/pmc4/$v -- python tools/pmc_decomp.py 100 1 > $R/gpurun_out/pmc4/$v.log 2>&1)
  tail -1 $R/gpurun_out/pmc4/$v.log
done
