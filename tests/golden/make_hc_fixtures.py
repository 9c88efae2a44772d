"""Blocks written by a DIFFERENT encoder — liblz4 1.9.3's LZ4_compress_HC (levels 9, 12) and LZ4_compress_fast
(acceleration 8) — as decode fixtures: optimal-parse streams have shapes lz-fear's greedy encoder never emits (long
matches chained back to back, many zero-literal sequences, matches chosen across overlapping candidates, offsets near
65535), and the decompressor must take them like the reference does.

  python tests/golden/make_hc_fixtures.py       (build container; needs liblz4.so.1)

Writes tests/golden/hc_blocks.bin (the compressed blocks, concatenated) and hc_blocks.json (per block: the generator call
that reproduces the input, offset/length inside the .bin, (length, xxh32) of input and of the compressed bytes)."""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle_ffi as o  # noqa: E402
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import synth  # noqa: E402

INPUTS = [   # (name, generator expression evaluated with `synth` in scope)
    ("mix_text", "synth.silesia_mix(0, 70000)"),
    ("mix_binary", "synth.silesia_mix(60 << 20, (60 << 20) + 70000)"),
    ("mix_db", "synth.silesia_mix(120 << 20, (120 << 20) + 66000)"),
    ("zipf", "synth.gen_text_zipf(77, 50000)"),
    ("log", "synth.log_text(0, 66000)"),
    ("repeat", "synth.repeat256(70000)"),
]


def main():
    L = C.CDLL("liblz4.so.1")
    L.LZ4_versionString.restype = C.c_char_p
    L.LZ4_compress_HC.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.LZ4_compress_fast.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    blob, meta = b"", {"liblz4": L.LZ4_versionString().decode(), "blocks": []}
    for name, expr in INPUTS:
        data = eval(expr, {"synth": synth}).tobytes()
        cap = len(data) + len(data) // 255 + 64
        for enc, arg in (("hc", 9), ("hc", 12), ("fast", 8)):
            out = C.create_string_buffer(cap)
            n = (L.LZ4_compress_HC if enc == "hc" else L.LZ4_compress_fast)(data, out, len(data), cap, arg)
            assert n > 0
            comp = out.raw[:n]
            rc, dec = o.decompress_raw(comp, limit=len(data))
            assert rc == 0 and dec == data            # the oracle agrees with the encoder's own input
            meta["blocks"].append({"name": f"{name}.{enc}{arg}", "input": expr, "offset": len(blob), "length": n,
                                   "in": [len(data), "%08x" % o.xxh32(data)], "comp": [n, "%08x" % o.xxh32(comp)]})
            blob += comp
    open(os.path.join(HERE, "hc_blocks.bin"), "wb").write(blob)
    json.dump(meta, open(os.path.join(HERE, "hc_blocks.json"), "w"), indent=1)
    print(len(meta["blocks"]), "blocks,", len(blob), "bytes")


if __name__ == "__main__":
    main()
