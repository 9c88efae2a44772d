#!/bin/bash
# kernels of the last 980-block call of bench.py's batch_sweep (distinct inputs), start / end in ms from the call's first kernel
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp
rm -rf /tmp/segb; (cd $R && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/segb -- python bench.py --no-cpu --no-e2e --no-config4 --no-config5 --no-verify --steps 1 --warmup 0 > /tmp/segb.log 2>&1)
f=$(ls /tmp/segb/*/*kernel_trace.csv | head -1); python - "$f" "${1:-32768}" <<'PY'
import csv,sys
ev=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name']
    if 'lzf' in n and 'compress_' not in n: ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),n.split('(')[0][-44:], r.get('Queue_Id','')))
ev.sort()
# the last call that used the 32 KiB ring
last=max(i for i,e in enumerate(ev) if ('resolve_pair_kernel<' + (sys.argv[2] if len(sys.argv) > 2 else '32768') + '>') in e[2])
# walk back to the call's first kernel: a gap of more than 0.5 ms in starts, or a plan kernel
i=last
while i>0 and not ('by_len' in ev[i][2]): i-=1
t0=ev[i][0]; end=max(e[1] for e in ev[i:last+1])
for s,e,n,q in ev[i:]:
    if s>end: break
    print(f"{(s-t0)/1e6:8.3f} .. {(e-t0)/1e6:8.3f}  q{q}  {n}")
print(f"call: {(end-t0)/1e6:.3f} ms")
PY
