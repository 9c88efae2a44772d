"""Randomised parity stress of the many-frames entry points (lzf_frame_compress_many / lzf_frame_decompress_many) against
the oracle's frame layer, not part of the test suite:
    python tests/stress_frames_many.py [rounds] [seed]
Each round: random settings shared by a batch of 8-40 random inputs (empty ones, stored blocks, streams of different
lengths), frames must equal the oracle's byte for byte; then the frames and damaged copies of them go through
decompress_many and must give the oracle's (status, bytes) each."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_ffi as o
import rust_lz_fear_amd
from rust_lz_fear_amd import framed
from stress_parity import make_input


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    rng = np.random.default_rng(seed)
    total = 0
    for r in range(rounds):
        bs = int(rng.choice([64 << 10, 64 << 10, 256 << 10, 1 << 20]))
        indep = bool(rng.integers(0, 2)); bsum = bool(rng.integers(0, 2)); csum = bool(rng.integers(0, 2))
        d = None
        if rng.integers(0, 2) == 0:
            d = make_input(rng, int(rng.choice([7, 200, 5000, 70000, 140000]))) or b"dictionary!"
        nf = int(rng.integers(8, 41))
        datas = [(make_input(rng, m) if m else b"") for m in (int(rng.choice([0, 11, 300, 70000, 200000, 600000])) for _ in range(nf))]
        es = o.make_settings(independent_blocks=indep, block_checksums=bsum, content_checksum=csum, block_size=bs,
                             dictionary=d, dictionary_id=(7 if d is not None else None))
        gs = framed.CompressionSettings().independent_blocks(indep).block_checksums(bsum).content_checksum(csum).block_size(bs)
        if d is not None:
            gs = gs.dictionary(7, d)
        frames = gs.compress_many(datas)
        for x, f in zip(datas, frames):
            erc, ef = o.frame_compress(x, es)
            assert erc == 0 and f == ef, ("frame bytes", r, len(x), bs, indep, bsum, csum, d is not None and len(d))
        allf = list(frames)
        for f in frames:
            b = bytearray(f)
            for _k in range(int(rng.integers(1, 3))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            allf.append(bytes(b))
        cap = 2 << 20
        got = framed.decompress_frames(allf, dictionary=d or b"", caps=[cap] * len(allf))
        for i, (f, (rc, out)) in enumerate(zip(allf, got)):
            erc, eout, _ = o.frame_decompress(f, dictionary=d or b"", cap=cap)
            assert rc == erc and out == eout, ("decode", r, i, rc, erc, len(out), len(eout))
        total += len(allf)
        print(f"round {r}: ok ({nf} frames, bs {bs}, indep {indep}, dict {len(d) if d else 0}; {total} decoded so far)", flush=True)
    print("stress ok")


if __name__ == "__main__":
    main()
