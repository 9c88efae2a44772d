"""Host-buffer frame layer timed end to end through the C ABI on one 202 MiB frame (measurement)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import framed, synth, ffi
data = synth.silesia_mix().tobytes()
f = framed.CompressionSettings().compress(data)
cap = len(data) + (8 << 20)
out = C.create_string_buffer(cap); n = C.c_size_t(0); used = C.c_size_t(0)
for it in range(3):
    t = time.time(); rc = ffi.lib().lzf_frame_decompress(f, len(f), b"", 0, out, cap, C.byref(n), C.byref(used)); dt = time.time() - t
    print(f"lzf_frame_decompress alone: rc {rc} {dt*1e3:.0f} ms -> {len(data)/dt/2**30:.2f} GiB/s")
s = framed.CompressionSettings()._struct(None)
bound = ffi.lib().lzf_frame_compress_bound(C.byref(s), len(data))
ob = C.create_string_buffer(bound); on = C.c_size_t(0)
for it in range(2):
    t = time.time(); rc = ffi.lib().lzf_frame_compress(C.byref(s), data, len(data), ob, bound, C.byref(on)); dt = time.time() - t
    print(f"lzf_frame_compress alone: rc {rc} {dt*1e3:.0f} ms -> {len(data)/dt/2**30:.2f} GiB/s")
t = time.time(); h = ffi.lib().lzf_xxh32(data, len(data), 0); print(f"host xxh32 of the content: {(time.time()-t)*1e3:.0f} ms")
