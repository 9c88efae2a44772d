import os, sys, time
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import framed, synth
data = synth.silesia_mix().tobytes()
for name, st in (("default (content checksum)", framed.CompressionSettings()), ("no content checksum", framed.CompressionSettings().content_checksum(False))):
    framed.CompressionSettings().compress(data[:1 << 20])
    t = time.time(); f = st.compress(data); tc = time.time() - t
    t = time.time(); back = framed.decompress_frame(f, cap=len(data) + (8 << 20)); td = time.time() - t
    assert back == data
    print(f"{name}: {len(data)/2**20:.0f} MiB -> {len(f)/2**20:.0f} MiB; compress {len(data)/tc/2**30:.2f} GiB/s, decompress {len(data)/td/2**30:.2f} GiB/s (host buffers, end to end)")
