#!/bin/bash
# round 6: wave-instructions per sequence of the bitmap-fed call's kernels (parse, seam, fed) — two PMC passes (instruction mix; wait states), one group per call
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export TMPDIR=/tmp; C=${1:-48}
export LZF_LIB_PATH="${LZF_LIB_PATH:-$R/rust-lz-fear_amd/liblzfear_hip_analysis.so}" LZF_FED_GROUPS=1 LZF_DECOMPRESS_KERNEL=fed
mkdir -p gpurun_out
{
for PASS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
rm -rf /tmp/fedc; timeout 900 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d /tmp/fedc -- python tools/pmc_decomp.py $C 1 > /tmp/fedc.log 2>&1
grep "^jobs" /tmp/fedc.log | tail -1
f=$(ls /tmp/fedc/*/*counter_collection.csv | head -1); python - "$f" "$C" <<'PY'
import csv, sys, collections
seqs = 11.71e6 * int(sys.argv[2])          # sequences of the corpus' 49 blocks x copies
agg = collections.defaultdict(collections.Counter); disp = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'lzf_seg' in n or 'paired' in n or 'fed' in n:
        k = n.split('(')[0][-48:]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k][r['Counter_Name']] += 1
print(f"per sequence ({seqs/1e6:.0f} M sequences), one dispatch each:")
for k, c in agg.items():
    print(f"  {k}")
    print("     " + "  ".join(f"{name[3:]} {v / disp[k][name] / seqs:.3f}" for name, v in sorted(c.items())))
PY
done
} > gpurun_out/fed_counters.log 2>&1
cat gpurun_out/fed_counters.log
