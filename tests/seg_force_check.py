"""The fall-backs of the segmented decompress pipeline, forced (analysis library, LZF_SEG_FORCE = noscratch | stager | resolver):
no scratch memory -> the pair kernel takes the whole batch; a stager / resolver wave of every odd job gives up in its third batch
(what a bounded wait that expires does) -> the job is NOT reported by the pipeline and the pair kernel decodes it from its first
byte.  Either way: statuses and bytes of the oracle.  Run as a script by tests/test_gpu_hardening.py (the knob is read once per process)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402
import oracle_ffi as o  # noqa: E402
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import ffi, synth  # noqa: E402


def main():
    force = os.environ.get("LZF_SEG_FORCE", "")
    mib = 1 << 20
    gens = [synth.gen_text_zipf, synth.gen_markup, synth.gen_exe, synth.gen_records, synth.gen_walk16, synth.gen_log]
    raws = [g(300 + i, mib + 999 * i).tobytes() for i, g in enumerate(gens)] + [synth.silesia_mix(5 * mib, 8 * mib).tobytes(),
            bytes(mib) + synth.gen_text_zipf(9, 200000).tobytes() + bytes(mib), synth.silesia_mix(40 * mib, 42 * mib).tobytes()]
    items, exp = [], []
    for d in raws:
        c = o.compress2(d)[1]
        assert len(c) >= 65536 or d[:8] == bytes(8)
        items.append(dict(input=c, limit=len(d), out_cap=len(d) + len(c) + 64)); exp.append((0, d))
    rng = np.random.default_rng(17)
    c0 = o.compress2(raws[0])[1]
    for k in range(3):                                       # damaged blocks among them: the pair kernel's statuses
        b = bytearray(c0)
        for _ in range(2 + k):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        m = bytes(b); cap = len(raws[0]) + len(m) + 64
        items.append(dict(input=m, limit=len(raws[0]), out_cap=cap)); exp.append(o.decompress_raw(m, limit=len(raws[0]), cap=cap))
    res = ffi.decompress_blocks_host(items)
    launch = ffi.lib().lzf_last_decompress_launch().decode()
    for i, ((rc, out), (erc, eout)) in enumerate(zip(res, exp)):
        assert rc == erc, (force, i, rc, erc)
        if rc == 0:
            assert out == eout, (force, i, "bytes differ")
    if force == "noscratch":
        assert not launch.startswith("segmented"), launch
    else:
        assert launch.startswith("segmented"), launch
    print("force ok:", force, launch)


if __name__ == "__main__":
    main()
