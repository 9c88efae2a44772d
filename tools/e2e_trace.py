"""End-to-end timing of the frame drivers from host buffers, with the phase trace of the analysis library
(LZF_LIB_PATH=rust-lz-fear_amd/liblzfear_hip_analysis.so LZF_FRAME_TRACE=1)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_lz_fear_amd  # noqa
from rust_lz_fear_amd import ffi, framed, synth

F = int(os.environ.get("E2E_FRAMES", "64")); fsz = int(os.environ.get("E2E_FRAME_MIB", "16")) << 20
bs = int(os.environ.get("E2E_BS", str(4 << 20)))
mix = synth.silesia_mix(0, 200 << 20)
datas = [mix[(i * fsz) % (mix.size - fsz):][:fsz].tobytes() for i in range(F)]
L = ffi.lib(); n = F; total = n * fsz
s = framed.CompressionSettings().block_size(bs)._struct(None)
caps = [L.lzf_frame_compress_bound(C.byref(s), len(d)) for d in datas]
outs = [C.create_string_buffer(c) for c in caps]
ins = (C.c_char_p * n)(*datas); lens = (C.c_size_t * n)(*[len(d) for d in datas])
outp = (C.c_void_p * n)(*[C.cast(o, C.c_void_p) for o in outs]); capa = (C.c_size_t * n)(*caps)
olen = (C.c_size_t * n)(); st = (C.c_int * n)()
for it in range(3):
    t = time.perf_counter(); rc = L.lzf_frame_compress_many(C.byref(s), n, ins, lens, outp, capa, olen, st); dt = time.perf_counter() - t
    print(f"compress_many call {it}: {dt*1e3:.1f} ms  {total/dt/2**30:.2f} GiB/s", file=sys.stderr)
frames = [outs[f].raw[: olen[f]] for f in range(n)]
dcap = [len(d) + 64 for d in datas]; douts = [C.create_string_buffer(c) for c in dcap]
fin = (C.c_char_p * n)(*frames); flen = (C.c_size_t * n)(*[len(f) for f in frames])
doutp = (C.c_void_p * n)(*[C.cast(o, C.c_void_p) for o in douts]); dcapa = (C.c_size_t * n)(*dcap)
dlen = (C.c_size_t * n)(); used = (C.c_size_t * n)(); dst = (C.c_int * n)()
for it in range(3):
    t = time.perf_counter(); rc = L.lzf_frame_decompress_many(n, fin, flen, None, 0, doutp, dcapa, dlen, used, dst); dt = time.perf_counter() - t
    print(f"decompress_many call {it}: {dt*1e3:.1f} ms  {total/dt/2**30:.2f} GiB/s", file=sys.stderr)
assert all(douts[f].raw[: dlen[f]] == datas[f] for f in range(0, n, 8))
print(ffi.frame_stats(), file=sys.stderr)
