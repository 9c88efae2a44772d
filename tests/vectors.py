"""Shared seeded test inputs (CPU-side; used by oracle tests and by the GPU parity tests)."""
import numpy as np

import rust_lz_fear_amd  # noqa: F401  (import shim)
from rust_lz_fear_amd import synth

# The round-trip strings of the reference's unit tests (src/lib.rs:43-95) — test *data*.
LIB_RS_STRINGS = [
    b"to live or not to live",
    b"Love is a wonderful terrible thing",
    b"There is nothing either good or bad, but thinking makes it so.",
    b"I burn, I pine, I perish.",
    b"To cute to die! Save the red panda!",
    b"You are 60% water. Save 60% of yourself!",
    b"Save water, it doesn't grow on trees.",
    b"The panda bear has an amazing black-and-white fur.",
    b"The average panda eats as much as 9 to 14 kg of bamboo shoots a day.",
    b"The Empress Dowager Bo was buried with a panda skull in her vault",
    b"as6yhol.;jrew5tyuikbfewedfyjltre22459ba",
    b"jhflkdjshaf9p8u89ybkvjsdbfkhvg4ut08yfrr",
    b"ahhd", b"ahd", b"x-29", b"x", b"k", b".", b"ajsdh", b"",
    b"\0" * 13,
    b"The Read trait allows for reading bytes from a source. Implementors of the Read trait are "
    b"called 'readers'. Readers are defined by one required method, read().",
]

# decode KATs of src/raw/decompress.rs:153-175: (input, expected status name, expected output)
DECODE_KATS = [
    (bytes([0x11, ord("a"), 1, 0]), 0, b"aaaaaa"),
    (bytes([0x11, ord("a"), 1, 0, 0x22, ord("b"), ord("c"), 2, 0]), 0, b"aaaaaabcbcbcbc"),
    (bytes([0x30, ord("a"), ord("4"), ord("9")]), 0, b"a49"),
    (bytes([0x10, ord("a"), 2, 0]), 4, None),   # offset_oob -> InvalidDeduplicationOffset
    (bytes([0x40, ord("a"), 1, 0]), 1, None),   # literal run past the end -> UnexpectedEnd
]


def big_compression_bytes(n):
    """src/lib.rs:98-106 generator: (n as u8)*10 + 33 ^ 0xA2."""
    k = np.arange(n, dtype=np.uint64)
    return (((k & 0xFF) * 10 + 33) & 0xFF ^ 0xA2).astype(np.uint8).tobytes()


def rng_bytes(seed, n):
    return synth.gen_random(seed, n).tobytes()


def small_cases():
    """Adversarially small / boundary inputs (the 12-byte MFLIMIT and 5-byte LASTLITERALS rules)."""
    cases = []
    for n in list(range(0, 40)) + [63, 64, 65, 66, 67, 127, 128, 129, 255, 256, 257, 300, 1000]:
        cases.append((f"zeros{n}", bytes(n)))
        cases.append((f"ab{n}", (b"ab" * n)[:n]))
        cases.append((f"rnd{n}", rng_bytes(1000 + n, n)))
        cases.append((f"text{n}", synth.gen_text_zipf(77, max(n, 1)).tobytes()[:n]))
    return cases


def medium_cases():
    cases = []
    for i, (name, _, cls, kw) in enumerate(synth.SILESIA_SEGMENTS):
        cases.append((f"{name}_256k", synth.CLASSES[cls](4242 + i, 256 << 10, **kw).tobytes()))
    cases.append(("log_512k", synth.gen_log(5, 512 << 10).tobytes()))
    cases.append(("repeat256_64k", synth.repeat256(65536).tobytes()))
    cases.append(("zeros_1m", bytes(1 << 20)))
    cases.append(("bench_shape_1m", bytes(200000) + rng_bytes(9, 400000) + bytes(400000)))  # benches/my_benchmark.rs:12-13 shape
    cases.append(("bigcomp_1m", big_compression_bytes(1 << 20)))
    cases.append(("lcg3_mask3_69632", synth.lcg_bytes(3, 69632, 3)))
    cases.append(("mixed_1m", synth.silesia_mix(9 << 20, 10 << 20).tobytes()))
    return cases


def lsic(v):
    """LSIC tail bytes of a length whose nibble already holds min(v, 15) (mod.rs:243-260)."""
    if v < 15:
        return b""
    v -= 15
    return b"\xff" * (v // 255) + bytes([v % 255])


def synth_stream(seed, n_seq, profile):
    """A VALID raw LZ4 block built sequence by sequence (not by a compressor), so that shapes no
    greedy parse would emit are covered: dense 3-byte tokens, huge literal runs and matches,
    offsets 1..8, offsets near 64 KiB, prefix-free.  Returns (block bytes, decoded bytes)."""
    rng = np.random.default_rng(seed)
    out = bytearray()
    blk = bytearray()
    for i in range(n_seq):
        r = rng.random()
        if profile == "dense":            # tokens every 3 bytes: no literals, 4-byte matches
            L = 0 if i else 8
            M = 4 + int(rng.integers(0, 3))
        elif profile == "mixed":
            L = int(rng.integers(0, 20)) if r < 0.9 else int(rng.integers(20, 400))
            M = 4 + (int(rng.integers(0, 24)) if r < 0.85 else int(rng.integers(24, 700)))
        elif profile == "long":
            L = int(rng.integers(0, 6000)) if r < 0.3 else int(rng.integers(0, 40))
            M = 4 + (int(rng.integers(0, 90000)) if r > 0.8 else int(rng.integers(0, 60)))
        elif profile == "classes":        # every copy path of the segmented pipeline's resolver: the four two-ended sizes, the
            L = int(rng.integers(0, 7)) if r < 0.95 else int(rng.integers(7, 200))       # same for run-length offsets, long
            b = rng.random()                                                              # and overlapping ones, bursts of 65..160
            M = (int(rng.integers(4, 8)) if b < 0.2 else int(rng.integers(8, 17)) if b < 0.4 else int(rng.integers(17, 33)) if b < 0.55 else
                 int(rng.integers(33, 65)) if b < 0.7 else int(rng.integers(65, 161)) if b < 0.85 else int(rng.integers(161, 700)) if b < 0.985 else
                 int(rng.integers(2000, 9000)))
        else:                             # "rle": tiny offsets, overlapping copies
            L = int(rng.integers(0, 5))
            M = 4 + int(rng.integers(0, 300))
        if len(out) + L == 0:
            L = 4
        lit = bytes(rng.integers(0, 256, L, dtype=np.uint8))
        avail = len(out) + L
        if profile == "classes":
            q = rng.random()
            if q < 0.4: off = int(rng.choice([1, 2, 4]))
            elif q < 0.55: off = int(rng.integers(1, 65))
            elif q < 0.8: off = int(rng.integers(200, 66000))          # (far: the same level for a run of them)
            else: off = int(rng.integers(1, 66000))
        elif profile == "rle" or rng.random() < 0.15:
            off = int(rng.integers(1, min(avail, 9) + 1)) if avail >= 1 else 1
        elif rng.random() < 0.1:
            off = min(avail, 65535 - int(rng.integers(0, 50)))
        else:
            off = int(rng.integers(1, min(avail, 65535) + 1))
        off = max(1, min(off, avail, 65535))
        blk += bytes([(min(L, 15) << 4) | min(M - 4, 15)]) + lsic(L) + lit + off.to_bytes(2, "little") + lsic(M - 4)
        out += lit
        start = len(out) - off
        for k in range(M):
            out.append(out[start + k])
    # last literals
    tail = bytes(rng.integers(0, 256, 7, dtype=np.uint8))
    blk += bytes([len(tail) << 4]) + tail
    out += tail
    return bytes(blk), bytes(out)
