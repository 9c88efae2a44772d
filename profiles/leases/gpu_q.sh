cd "$GRAFT_REPO_ROOT"
for L in liblzfear_hip.so liblzfear_hip_e2348aca72.so; do echo -n "$L: "; LZF_LIB_PATH=$PWD/rust-lz-fear_amd/$L LZF_VERIFY=1 timeout 300 python tools/pmc_decomp.py 240 3 2>&1 | tail -2 | tr '\n' ' '; echo; done
