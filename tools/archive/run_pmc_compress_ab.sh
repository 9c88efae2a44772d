set -u
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for lib in old new; do
for b in ${LZF_PMC_BLOCKS:-16 25 28 47 49}; do
rm -rf $R/gpurun_out/pmc7
if [ $lib = old ]; then export LZF_LIB_PATH=dbg/lib_old.so; else unset LZF_LIB_PATH; fi
(cd $R && LZF_ONLY_BLOCK=$b timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc7 -- python tools/time_compress.py 400 1 > $R/gpurun_out/pmc7.log 2>&1)
f=$(ls $R/gpurun_out/pmc7/*/*_counter_collection.csv | head -1)
python - "$f" $b $lib <<PY
import csv,sys,collections
agg=collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "compact_kernel<false>" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
print(sys.argv[3],"block",sys.argv[2],{k:round(v/400/1e6,2) for k,v in agg.items()})
PY
done; done
