# kernel-variant knobs live in the analysis flavour of the library (rust-lz-fear_amd/build.py)
export LZF_LIB_PATH="${LZF_LIB_PATH:-${GRAFT_REPO_ROOT:-$PWD}/rust-lz-fear_amd/liblzfear_hip_analysis.so}"
for v in v6l256; do echo "== $v"; LZF_DECOMPRESS_KERNEL=$v timeout 300 python tests/variant_check.py 2>&1 | tail -1; done
echo "== stress v6l256"; LZF_DECOMPRESS_KERNEL=v6l256 timeout 600 python tests/stress_parity.py 2 95 2>&1 | tail -1
bash tools/time_variants.sh 240 paired24 v6l256 v6l256w6
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/v6stats
(cd $R && LZF_DECOMPRESS_KERNEL=v6l256 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v6stats -- python tools/pmc_decomp.py 240 2 > $R/gpurun_out/v6stats.log 2>&1)
python - <<'PY'
import csv,glob,os
R=os.environ['GRAFT_REPO_ROOT']
for f in glob.glob(R+'/gpurun_out/v6stats/**/*kernel_stats.csv',recursive=True):
    for r in list(csv.DictReader(open(f)))[:5]:
        if 'v6' in r['Name']: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e6,'ms')
PY
rm -rf $R/gpurun_out/v6stats
