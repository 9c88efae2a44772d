"""Randomised frame-level parity stress (GPU frame layer vs the oracle's), not part of the test suite:
    python tests/stress_frames.py [rounds] [seed]
Random CompressionSettings (block size, linked / independent, checksums, dictionary, content size) on random inputs:
the GPU frame must equal the oracle's frame byte for byte, decode back to the input, and damaged frames must fail
(or succeed) with the oracle's status."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_ffi as o
import rust_lz_fear_amd
from rust_lz_fear_amd import framed, synth
from stress_parity import make_input


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rng = np.random.default_rng(seed)
    n = 0
    for r in range(rounds):
        for _ in range(10):
            data = make_input(rng, int(rng.choice([300, 70000, 300000, 1200000])))
            bs = int(rng.choice([64 << 10, 256 << 10, 1 << 20, 4 << 20]))
            indep = bool(rng.integers(0, 2)); bsum = bool(rng.integers(0, 2)); csum = bool(rng.integers(0, 2))
            d = None
            if rng.integers(0, 3) == 0:
                d = make_input(rng, int(rng.choice([7, 200, 5000, 70000, 140000]))) or b"dictionary!"
            with_size = bool(rng.integers(0, 2))
            es = o.make_settings(independent_blocks=indep, block_checksums=bsum, content_checksum=csum, block_size=bs,
                                 dictionary=d, dictionary_id=(7 if d is not None else None),
                                 content_size=(len(data) if with_size else None))
            erc, eframe = o.frame_compress(data, es)
            gs = framed.CompressionSettings().independent_blocks(indep).block_checksums(bsum).content_checksum(csum).block_size(bs)
            if d is not None:
                gs = gs.dictionary(7, d)
            gframe = gs.compress_with_size(data) if with_size else gs.compress(data)
            assert erc == 0 and gframe == eframe, ("frame bytes", r, len(data), bs, indep, bsum, csum, d is not None and len(d))
            assert framed.decompress_frame(gframe, dictionary=d or b"") == data, ("round trip", r)
            # a damaged copy
            b = bytearray(gframe)
            for _k in range(int(rng.integers(1, 3))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            m = bytes(b)
            erc2, eout, _ = o.frame_decompress(m, dictionary=d or b"", cap=max(len(data) * 2, 1 << 20) + (8 << 20))
            try:
                gout = framed.decompress_frame(m, dictionary=d or b"", cap=max(len(data) * 2, 1 << 20) + (8 << 20)); grc = 0
            except framed.FrameError as e:
                grc = e.code; gout = None
            assert grc == erc2 and (grc != 0 or gout == eout), ("damaged frame", r, grc, erc2)
            n += 1
        print(f"round {r}: ok ({n} frames)", flush=True)
    print("stress ok")


if __name__ == "__main__":
    main()
