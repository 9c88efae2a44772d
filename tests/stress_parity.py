"""Randomised GPU-vs-oracle parity stress (a script next to the test suite, not collected by pytest; run on a GPU box):
    python tests/stress_parity.py [rounds] [seed]
Every round builds a few dozen inputs out of random pieces (text, markup, exe-like, records, 16-bit walks, noise,
zero / motif runs of up to several 64 KiB epochs, copies of earlier pieces at random distances), compresses them on
the GPU and with the oracle (bytes must be equal, U32 and U16 tables, cursor > 0), decompresses with the kernel variant
LZF_DECOMPRESS_KERNEL selects (must reproduce the input), and decodes randomly damaged blocks (same status as the oracle).
"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_ffi as o
import rust_lz_fear_amd
from rust_lz_fear_amd import ffi, synth

GENS = [synth.gen_text_zipf, synth.gen_markup, synth.gen_exe, synth.gen_records, synth.gen_walk16, synth.gen_random, synth.gen_log]


def make_input(rng, max_len):
    parts, total = [], 0
    target = int(rng.integers(0, max_len))
    while total < target:
        kind = rng.integers(0, 10)
        n = int(min(target - total, rng.choice([7, 40, 300, 3000, 20000, 70000, 150000, 300000])))
        n = max(n, 1)
        if kind < 7:
            p = GENS[kind](int(rng.integers(1, 1 << 30)), n).tobytes()
        elif kind == 7:
            p = bytes([int(rng.integers(0, 256))]) * n
        elif kind == 8:
            m = rng.integers(0, 256, int(rng.integers(2, 300)), dtype=np.uint8).tobytes()
            p = (m * (n // len(m) + 1))[:n]
        else:
            if not parts:
                continue
            whole = b"".join(parts)
            a = int(rng.integers(0, len(whole))); p = whole[a:a + n]
        parts.append(p); total += len(p)
    return b"".join(parts)[:target]


def one_round(rng, r, sizes=(200, 5000, 70000, 400000, 1500000), n_inputs=24):
    """One round of the stress: returns (inputs checked, damaged blocks checked); raises AssertionError on any difference."""
    n_in = n_bad = 0
    if True:
        data = [make_input(rng, int(rng.choice(list(sizes)))) for _ in range(n_inputs)]
        # ---- compress, U32 fresh table
        res = ffi.compress_blocks_host([dict(input=d, out_cap=len(d) + len(d) // 200 + 64) for d in data])
        comps = []
        for d, (rc, out) in zip(data, res):
            erc, eout = o.compress2(d)
            assert rc == erc and out == eout, ("compress u32", r, len(d))
            comps.append(eout)
        # ---- U16 table (inputs < 64 KiB) and cursor > 0
        items, exp = [], []
        for d in data:
            if 0 < len(d) <= 65535:
                items.append(dict(input=d, kind=ffi.TABLE_U16)); exp.append(o.compress2(d, kind=o.TABLE_U16))
            if len(d) > 10:
                cur = int(rng.integers(1, len(d)))
                cap = int(rng.choice([len(d), max(1, len(d) // 3)]))
                items.append(dict(input=d, cursor=cur, out_cap=cap)); exp.append(o.compress2(d, cursor=cur, cap=cap))
        for it, (rc, out), (erc, eout) in zip(items, ffi.compress_blocks_host(items), exp):
            assert rc == erc and (rc != 0 or out == eout), ("compress variants", r, len(it["input"]), it.get("cursor"), it.get("kind"))
        # ---- decompress
        res = ffi.decompress_blocks_host([dict(input=c, limit=max(len(d), 1), out_cap=len(d) + len(c) + 64) for c, d in zip(comps, data)])
        for d, (rc, out) in zip(data, res):
            assert rc == 0 and out == d, ("decompress", r, len(d))
        # ---- damaged blocks: same status (and same bytes when Ok)
        items, exp = [], []
        for d, c in zip(data, comps):
            if not (0 < len(c) <= 400000):
                continue
            b = bytearray(c)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            if rng.integers(0, 3) == 0 and len(b) > 2:
                del b[int(rng.integers(1, len(b))):]
            m = bytes(b); limit = len(d); cap = limit + len(m) + 64
            exp.append(o.decompress_raw(m, limit=limit, cap=cap)); items.append(dict(input=m, limit=limit, out_cap=cap))
        for (rc, out), (erc, eout) in zip(ffi.decompress_blocks_host(items), exp):
            assert rc == erc and (rc != 0 or out == eout), ("damaged", r)
            n_bad += 1
        n_in += len(data)
    return n_in, n_bad


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    n_in = n_bad = 0
    for r in range(rounds):
        a, b = one_round(rng, r)
        n_in += a; n_bad += b
        print(f"round {r}: ok ({n_in} inputs, {n_bad} damaged blocks so far)", flush=True)
    print("stress ok")


if __name__ == "__main__":
    main()
