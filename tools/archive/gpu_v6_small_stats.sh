#!/bin/bash
export LZF_LIB_PATH="${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so"
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in v6l256 v6l128 v6s512; do
  rm -rf /tmp/v6s; (cd $R && LZF_DECOMPRESS_KERNEL=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/v6s -- python tools/pmc_decomp.py 4 3 > /tmp/v6s.log 2>&1)
  echo "== $v"; grep "^jobs" /tmp/v6s.log | tail -1
  f=$(ls /tmp/v6s/*/*kernel_stats.csv | head -1); python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'lzf' in r['Name']: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e6, 'ms')
PY
done
