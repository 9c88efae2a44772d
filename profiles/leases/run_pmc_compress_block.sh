set -u
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for b in ${LZF_PMC_BLOCKS:-0 4 20}; do
rm -rf $R/gpurun_out/pmc6
(cd $R && LZF_ONLY_BLOCK=$b timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc6 -- python tools/time_compress.py 1000 1 > $R/gpurun_out/pmc6.log 2>&1)
f=$(ls $R/gpurun_out/pmc6/*/*_counter_collection.csv | head -1)
python - "$f" $b <<PY
import csv,sys,collections
agg=collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "compact_kernel<false>" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
print("block",sys.argv[2],{k:v/1000 for k,v in agg.items()})
PY
done
