#!/bin/bash
# active lanes per match round of the headline decompress kernel (paired24): seven -DLZF_DBG_ROUNDS=k builds, one counter each
# (results[].reserved summed over the 49 compressible blocks of one corpus copy)
mkdir -p gpurun_out; out=gpurun_out/r04_lane_histogram.txt; rm -f $out
names=(x x "batches" "lanes moved in all solo rounds" "solo rounds" "solo rounds with 1 lane" "solo rounds with 2-4 lanes" "solo rounds with 5-16 lanes" "solo rounds with 17-64 lanes")
for k in 2 3 4 5 6 7 8; do
  r=$(LZF_LIB_PATH=rust-lz-fear_amd/liblzfear_hip_dbg_rounds$k.so LZF_DECOMPRESS_KERNEL=paired24 LZF_PRINT_RESERVED=1 timeout 300 python tools/pmc_decomp.py 1 1 2>&1 | grep "^reserved" | sed 's/.*sum \([0-9]*\).*/\1/')
  echo "${names[$k]}: $r" >> $out
done
cat $out
