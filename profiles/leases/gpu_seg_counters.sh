#!/bin/bash
# wave-instructions per sequence of the segmented pipeline's kernels (one PMC pass, one group per call so that every kernel is one dispatch)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
export LZF_SEG_MIN_IN=65536 LZF_SEG_GROUPS=100 LZF_DECOMPRESS_KERNEL=seg
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp; C=${1:-20}
rm -rf /tmp/segc; (cd $R && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/segc -- python tools/pmc_decomp.py $C 1 > /tmp/segc.log 2>&1)
grep "^jobs" /tmp/segc.log | tail -1
f=$(ls /tmp/segc/*/*counter_collection.csv | head -1); python - "$f" "$C" <<'PY'
import csv, sys, collections
seqs = 11.71e6 * int(sys.argv[2])          # sequences of the corpus' 49 blocks x copies
agg = collections.defaultdict(lambda: collections.Counter()); disp = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'lzf_seg' in n or 'paired' in n:
        k = n.split('(')[0][-48:]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k] += 1
print(f"{'kernel':50s} dispatches  VALU   SALU   LDS   VMEM   total   (wave-instructions per sequence, {seqs/1e6:.0f} M sequences)")
tot = 0
for k, c in agg.items():
    nd = disp[k] / 5          # five counters -> five rows per dispatch
    v, s_, l, m = c['SQ_INSTS_VALU'] / nd / seqs, c['SQ_INSTS_SALU'] / nd / seqs, c['SQ_INSTS_LDS'] / nd / seqs, (c['SQ_INSTS_VMEM_RD'] + c['SQ_INSTS_VMEM_WR']) / nd / seqs
    print(f"{k:50s} {nd:6.0f}   {v:6.2f} {s_:6.2f} {l:5.2f} {m:6.2f} {v + s_ + l + m:7.2f}"); tot += v + s_ + l + m
print(f"sum {tot:.2f}")
PY
