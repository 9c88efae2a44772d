#!/bin/bash
# round 6, review item 7: a linked-blocks call of 200 streams (64 KiB blocks, 2 MiB each): the team class for carried tables against the
# general kernel's lone wavefronts (LZF_COMPRESS_TEAM_MAX=0), same box; frames compared with each other
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
A=$PWD/rust-lz-fear_amd/liblzfear_hip_analysis.so
cat > /tmp/linked_team.py <<'PY'
import sys, os, time, hashlib
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import framed, synth, ffi
S = 200
datas = [synth.silesia_mix((k * 1000003) % (180 << 20), (k * 1000003) % (180 << 20) + (2 << 20)).tobytes() for k in range(S)]
cs = framed.CompressionSettings().block_size(64 << 10).independent_blocks(False)
ts = []
for _ in range(5):
    t = time.perf_counter(); out = cs.compress_many(datas); ts.append(time.perf_counter() - t)
h = hashlib.sha1(b"".join(out)).hexdigest()
print(f"{os.environ.get('LZF_COMPRESS_TEAM_MAX', 'default')}: lzf_frame_compress_many, {S} linked streams x 2 MiB, 64 KiB blocks: median {sorted(ts[1:])[2] * 1e3:.1f} ms ({S * 2 / 1024 / sorted(ts[1:])[2]:.2f} GiB/s); launch {ffi.lib().lzf_last_compress_launch().decode()}; frames sha1 {h[:12]}")
PY
{
LZF_LIB_PATH=$A timeout 600 python /tmp/linked_team.py 2>&1 | tail -1
LZF_LIB_PATH=$A LZF_COMPRESS_TEAM_MAX=0 timeout 600 python /tmp/linked_team.py 2>&1 | tail -1
} > gpurun_out/linked_team.log 2>&1
cat gpurun_out/linked_team.log
