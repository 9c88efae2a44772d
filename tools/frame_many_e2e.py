"""Many frames per call through the C ABI, host buffers in and out (measurement):
  (a) F frames of 16 MiB (4 MiB independent blocks, default settings)
  (b) BASELINE configs[4] shape: S linked streams of 4 MiB, 64 KiB blocks, 64 KiB motif dictionary
usage: python tools/frame_many_e2e.py [F] [S]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import rust_lz_fear_amd
from rust_lz_fear_amd import framed, synth, ffi
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L = ffi.lib()


def run_many(g, datas, dictionary=b"", reps=2, label=""):
    n = len(datas); total = sum(len(d) for d in datas)
    s = g._struct(None)
    caps = [L.lzf_frame_compress_bound(C.byref(s), len(d)) for d in datas]
    outs = [C.create_string_buffer(c) for c in caps]
    ins = (C.c_char_p * n)(*datas); lens = (C.c_size_t * n)(*[len(d) for d in datas])
    outp = (C.c_void_p * n)(*[C.cast(o, C.c_void_p) for o in outs]); capa = (C.c_size_t * n)(*caps)
    olen = (C.c_size_t * n)(); st = (C.c_int * n)()
    for it in range(reps):
        t = time.time(); rc = L.lzf_frame_compress_many(C.byref(s), n, ins, lens, outp, capa, olen, st); dt = time.time() - t
        print(f"{label} lzf_frame_compress_many: {n} frames, {total/2**20:.0f} MiB, rc {rc} bad {sum(1 for x in st if x)}: {dt*1e3:.0f} ms -> {total/dt/2**30:.2f} GiB/s", flush=True)
    frames = [outs[f].raw[: olen[f]] for f in range(n)]
    dcap = [len(d) + 64 for d in datas]
    douts = [C.create_string_buffer(c) for c in dcap]
    fin = (C.c_char_p * n)(*frames); flen = (C.c_size_t * n)(*[len(f) for f in frames])
    doutp = (C.c_void_p * n)(*[C.cast(o, C.c_void_p) for o in douts]); dcapa = (C.c_size_t * n)(*dcap)
    dlen = (C.c_size_t * n)(); used = (C.c_size_t * n)(); dst = (C.c_int * n)()
    for it in range(reps):
        t = time.time(); rc = L.lzf_frame_decompress_many(n, fin, flen, dictionary, len(dictionary), doutp, dcapa, dlen, used, dst); dt = time.time() - t
        print(f"{label} lzf_frame_decompress_many: rc {rc} bad {sum(1 for x in dst if x)}: {dt*1e3:.0f} ms -> {total/dt/2**30:.2f} GiB/s", flush=True)
    assert all(douts[f].raw[: dlen[f]] == datas[f] for f in range(0, n, max(1, n // 8)))
    # the same work one frame per call
    t = time.time()
    for f in range(min(n, 4)):
        on = C.c_size_t(0); L.lzf_frame_compress(C.byref(s), datas[f], len(datas[f]), outs[f], caps[f], C.byref(on))
    dt = time.time() - t; sub = sum(len(d) for d in datas[:4])
    print(f"{label} lzf_frame_compress, one frame per call ({min(n,4)} frames): {sub/dt/2**30:.3f} GiB/s", flush=True)
    t = time.time()
    for f in range(min(n, 4)):
        a = C.c_size_t(0); b = C.c_size_t(0); L.lzf_frame_decompress(frames[f], len(frames[f]), dictionary, len(dictionary), douts[f], dcap[f], C.byref(a), C.byref(b))
    dt = time.time() - t
    print(f"{label} lzf_frame_decompress, one frame per call: {sub/dt/2**30:.3f} GiB/s", flush=True)


mix = synth.silesia_mix().tobytes()
datas = [mix[(i * (16 << 20)) % (len(mix) - (16 << 20)):][: 16 << 20] for i in range(F)]
run_many(framed.CompressionSettings(), datas, label="(a)")
motif = synth.repeat256(65536).tobytes()
logt = synth.log_text(0, 4 << 20).tobytes()
streams = [(synth.repeat256((4 << 20) + i).tobytes()[i:] if i % 2 == 0 else logt[i:] + logt[:i]) for i in range(S)]
g = framed.CompressionSettings().block_size(64 << 10).independent_blocks(False).dictionary(5, motif)
run_many(g, streams, dictionary=motif, label="(b)")
