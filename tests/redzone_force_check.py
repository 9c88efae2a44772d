"""The red-zone check over the forced fall-backs of the segmented pipeline (analysis library, LZF_SEG_FORCE = noscratch | stager |
resolver; see tests/seg_force_check.py).  Run as a script by tests/test_gpu_redzone.py (the knob is read once per process)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402
import oracle_ffi as o  # noqa: E402
import redzone  # noqa: E402
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import synth  # noqa: E402


def main():
    mib = 1 << 20
    gens = [synth.gen_text_zipf, synth.gen_markup, synth.gen_exe, synth.gen_records, synth.gen_walk16, synth.gen_log]
    raws = [g(300 + i, mib + 999 * i).tobytes() for i, g in enumerate(gens)] + [synth.silesia_mix(5 * mib, 8 * mib).tobytes()]
    items, exp = [], []
    for d in raws:
        c = o.compress2(d)[1]
        items.append(dict(input=c, limit=len(d), out_cap=len(d))); exp.append((0, d))
    rng = np.random.default_rng(17)
    c0 = o.compress2(raws[0])[1]
    for k in range(4):
        b = bytearray(c0)
        for _ in range(2 + k):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        if k == 3:
            del b[len(b) // 2:]
        m = bytes(b); cap = len(raws[0]) + len(m) + 64
        items.append(dict(input=m, limit=len(raws[0]), out_cap=cap)); exp.append(o.decompress_raw(m, limit=len(raws[0]), cap=cap))
    redzone.check_decompress(items, exp, "forced " + os.environ.get("LZF_SEG_FORCE", ""))
    print("redzone force ok:", os.environ.get("LZF_SEG_FORCE", ""))


if __name__ == "__main__":
    main()
