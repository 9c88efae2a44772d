#!/bin/bash
# round 6: effective shader clock of the decompress kernels (GRBM_GUI_ACTIVE / kernel duration), 240 copies
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export TMPDIR=/tmp; C=${1:-240}
export LZF_LIB_PATH="$R/rust-lz-fear_amd/liblzfear_hip_analysis.so" LZF_FED_GROUPS=1
mkdir -p gpurun_out
{
for K in fed nofed; do
for P in 0 3328; do
rm -rf /tmp/fedk; LZF_DECOMPRESS_KERNEL=$K LZF_FED_PAD_LDS=$P timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d /tmp/fedk -- python tools/pmc_decomp.py $C 2 > /tmp/fedk.log 2>&1
echo "== $K pad $P: $(grep '^jobs' /tmp/fedk.log | tail -1)"
python - <<'PY'
import csv, glob, collections
cc = glob.glob('/tmp/fedk/*/*counter_collection.csv')[0]; kt = glob.glob('/tmp/fedk/*/*kernel_trace.csv')[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
seen = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    n = r['Kernel_Name']
    if 'fed_kernel' in n or 'paired' in n or 'seg_parse' in n:
        seen[(n.split('(')[0][-40:], r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
for (k, d), c in seen.items():
    ms = dur.get(d, 0)
    if ms > 0.5:
        print(f"  {k:42s} {ms:8.2f} ms  GRBM_GUI_ACTIVE {c.get('GRBM_GUI_ACTIVE', 0) / 1e6:8.1f} M  -> {c.get('GRBM_GUI_ACTIVE', 0) / ms / 1e6:6.3f} GHz   waves {c.get('SQ_WAVES', 0):.0f}")
PY
done
done
} > gpurun_out/fed_clock.log 2>&1
cat gpurun_out/fed_clock.log
