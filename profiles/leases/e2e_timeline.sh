#!/bin/bash
# timeline of the last lzf_frame_decompress_many call of tools/e2e_trace.py: kernels and PCIe copies (rocprofv3 traces)
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/tl
(cd $R && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -- python tools/e2e_trace.py > /tmp/tl.log 2>&1)
grep "decompress_many call" /tmp/tl.log
python - <<'PY'
import csv, glob, collections
kt = glob.glob('/tmp/tl/*/*kernel_trace.csv')[0]; mt = glob.glob('/tmp/tl/*/*memory_copy_trace.csv')[0]
ev = []
for r in csv.DictReader(open(kt)):
    n = r['Kernel_Name']
    if 'lzf' in n: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + n.split('(')[0][-44:]))
    elif 'copyBuffer' in n: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C D2H(blit-kernel) 4194304'))
rows = list(csv.DictReader(open(mt)))
for r in rows:
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r.get('Direction', r.get('Name', '?')) + ' ' + '4194304'))
ev.sort()
# the last call: starts at the last seg_plan kernel minus the uploads before it; take events after the last gap > 20 ms
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 50):i]) > 15_000_000: cut = i
ev = ev[cut:]; t0 = ev[0][0]
agg = []
for s, e, n in ev:
    if n.startswith('C '):
        d = n.split()[1]
        try: b = int(n.split()[2])
        except Exception: b = 0
        if agg and agg[-1][2] == 'C ' + d and s - agg[-1][1] < 300_000: agg[-1] = (agg[-1][0], e, 'C ' + d, agg[-1][3] + b, agg[-1][4] + 1)
        else: agg.append((s, e, 'C ' + d, b, 1))
    else: agg.append((s, e, n, 0, 1))
for s, e, n, b, c in agg:
    if n.startswith('C ') or (e - s) > 100_000:
        print(f"{(s - t0) / 1e6:8.3f} .. {(e - t0) / 1e6:8.3f} ms  {n}" + (f"  {b / 2**20:.0f} MiB in {c} copies, {b / max(1, e - s):.1f} GB/s" if b else ""))
PY
