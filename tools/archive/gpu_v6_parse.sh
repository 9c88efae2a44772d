# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
export LZF_V6_SLICE=16384
for v in v6l256 v6l128; do echo "== $v"; LZF_DECOMPRESS_KERNEL=$v timeout 600 python tests/variant_check.py 2>&1 | tail -1; done
echo "== stress v6l256"; LZF_DECOMPRESS_KERNEL=v6l256 timeout 900 python tests/stress_parity.py 3 71 2>&1 | tail -1
echo "== stress v6l128"; LZF_DECOMPRESS_KERNEL=v6l128 timeout 900 python tests/stress_parity.py 2 72 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in v6l128 v6l256 v6l384; do
rm -rf $R/gpurun_out/v6stats
(cd $R && LZF_DECOMPRESS_KERNEL=$v rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v6stats -- python tools/pmc_decomp.py 240 2 > $R/gpurun_out/v6stats.log 2>&1)
echo "$v: $(grep GiB $R/gpurun_out/v6stats.log | tail -1)"
python - <<'PY'
import csv,glob,os
R=os.environ['GRAFT_REPO_ROOT']
for f in glob.glob(R+'/gpurun_out/v6stats/**/*kernel_stats.csv',recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        if 'v6' in r['Name']: print('  ', r['Name'][:48], r['Calls'], float(r['AverageNs'])/1e6, 'ms')
PY
done
rm -rf $R/gpurun_out/v6stats
