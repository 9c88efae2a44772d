"""Which path the compact compress kernel takes, per block of the corpus (analysis build -DLZF_DBG_PATHS:
LZF_LIB_PATH=dbg/liblzf_paths.so).  Prints sequences by path and batches by kind."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_lz_fear_amd  # noqa
from rust_lz_fear_amd import ffi, synth
import ctypes as C

BS = 4 << 20
mix = synth.silesia_mix()
blocks = [mix[i:i + BS].tobytes() for i in range(0, mix.size, BS)]
n = len(blocks)
jobs = (ffi.CompressJob * n)(); res = (ffi.JobResult * n)(); keep = []
for i, b in enumerate(blocks):
    ib = C.create_string_buffer(b, len(b)); ob = C.create_string_buffer(len(b))
    keep.append((ib, ob))
    jobs[i].input = C.cast(ib, C.c_void_p); jobs[i].input_len = len(b); jobs[i].out = C.cast(ob, C.c_void_p); jobs[i].out_cap = len(b)
    jobs[i].table_kind = ffi.TABLE_U32
ffi.check(ffi.lib().lzf_compress_batch_host(jobs, res, n))
tot = np.zeros(6, dtype=np.int64)
names = ["straight", "ext_straight", "tail_short", "tail_long", "batches_fast", "batches_general"]
for i in range(n):
    if res[i].status == 0 and res[i].out_len >= 24:
        c = np.frombuffer(keep[i][1].raw[:24], dtype=np.uint32).astype(np.int64)
        tot += c
        if i % 6 == 0:
            print(i, dict(zip(names, c.tolist())))
seqs = tot[:4].sum()
print("total", dict(zip(names, tot.tolist())), "sequences", int(seqs))
print("share of sequences:", {k: round(float(v) / seqs, 4) for k, v in zip(names[:4], tot[:4])}, "batches per sequence: fast %.3f general %.3f" % (tot[4] / seqs, tot[5] / seqs))
