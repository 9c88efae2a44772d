# SQ instruction counters of the compress kernel (20 copies): bash profiles/leases/run_pmc_compress.sh  (through gpurun)
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc5; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmc5/c -- python tools/time_compress.py 20 1 > $R/gpurun_out/pmc5/c.log 2>&1)
tail -2 $R/gpurun_out/pmc5/c.log
