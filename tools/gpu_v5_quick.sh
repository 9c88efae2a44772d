for v in v5s512 v5s256 v5s1024; do echo "== $v"; LZF_DECOMPRESS_KERNEL=$v timeout 600 python tests/variant_check.py 2>&1 | tail -2; done
echo "== stress v5s512"; LZF_DECOMPRESS_KERNEL=v5s512 timeout 900 python tests/stress_parity.py 2 21 2>&1 | tail -2
bash tools/time_variants.sh 240 paired24 v5s512 v5s512w6 v5s256 v5s1024
bash tools/pmc_libs.sh v5s512 rust-lz-fear_amd/liblzfear_hip.so dbg/lib_s2_1.so dbg/lib_s2_32.so
