# TA / TD / TCP counters of the decompress kernel in separate rocprofv3 --pmc passes: bash profiles/leases/run_pmc_tcp.sh [lib.so]
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc3; cd /tmp; export TMPDIR=/tmp
LIB=${1:-$R/dbg/lib_full.so}
i=0
for grp in "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCP_TOTAL_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  (cd $R && LZF_LIB_PATH=$LIB timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc3/g$i -- python tools/pmc_decomp.py 100 1 > $R/gpurun_out/pmc3/g$i.log 2>&1)
  tail -1 $R/gpurun_out/pmc3/g$i.log
done
