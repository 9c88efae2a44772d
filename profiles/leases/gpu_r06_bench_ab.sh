#!/bin/bash
# round 6: A/B in bench.py's own setting (distinct inputs per copy) of libraries and environment knobs: each argument is "ENV=.. ENV=.. lib.so"
mkdir -p gpurun_out; rm -f gpurun_out/bench_ab.log
for rep in 1 2; do
for A in "$@"; do
  L=${A##* }; E=${A% *}; [ "$E" = "$A" ] && E=""
  env $E LZF_LIB_PATH=$PWD/rust-lz-fear_amd/$L timeout 900 python bench.py --no-cpu --no-e2e --no-config4 --no-config5 --no-verify --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print('$A', '| decompress', l['value'], 'ms', l['ms_per_step'], '| sweep', [v['ms'] for v in (l.get('batch_sweep') or {}).values()])" >> gpurun_out/bench_ab.log
done; done
cat gpurun_out/bench_ab.log
