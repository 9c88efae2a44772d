"""Reduced parity check of whatever decompress kernel generation LZF_DECOMPRESS_KERNEL selects.
Run as a script by tests/test_gpu_parity.py::test_every_decompress_kernel_generation (the choice is
read once per process, hence the subprocess)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402
import oracle_ffi as o  # noqa: E402
import vectors  # noqa: E402
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import ffi  # noqa: E402


def main():
    cases = vectors.small_cases()[::3] + vectors.medium_cases()
    comps = [o.compress2(d)[1] for _, d in cases]
    res = ffi.decompress_blocks_host([dict(input=c, limit=max(len(d), 1), out_cap=len(d) + len(c) + 64) for c, (_, d) in zip(comps, cases)])
    for (name, d), (rc, out) in zip(cases, res):
        assert rc == 0 and out == d, name
    # malformed inputs: same error kind as the oracle
    rng = np.random.default_rng(99)
    items, exp = [], []
    for (name, d), c in zip(cases, comps):
        if not (0 < len(d) <= 300000):
            continue
        for k in range(3):
            b = bytearray(c)
            for _ in range(1 + k):
                if len(b):
                    i = rng.integers(0, len(b)); b[i] = rng.integers(0, 256)
            if k == 2 and len(b) > 2:
                del b[rng.integers(1, len(b)):]
            m = bytes(b)
            limit = len(d); cap = limit + len(m) + 64
            exp.append(o.decompress_raw(m, limit=limit, cap=cap))
            items.append(dict(input=m, limit=limit, out_cap=cap))
    res = ffi.decompress_blocks_host(items)
    for (erc, eout), (rc, out) in zip(exp, res):
        assert rc == erc
        if rc == 0:
            assert out == eout
    # prefix / existing output
    d = vectors.synth.gen_text_zipf(31, 50000).tobytes()
    dic, payload = d[:20000], d[20000:]
    comp = o.compress2(dic + payload, cursor=len(dic))[1]
    r = ffi.decompress_blocks_host([dict(input=comp, prefix=dic, limit=len(payload)), dict(input=comp, existing=dic, limit=len(d))])
    assert r[0] == (0, payload) and r[1] == (0, d)
    # handcrafted streams (rare kernel paths)
    for seed, n, prof in [(11, 4000, "dense"), (12, 2000, "mixed"), (13, 200, "long"), (14, 1500, "rle")]:
        blk, out = vectors.synth_stream(seed, n, prof)
        (rc, got), = ffi.decompress_blocks_host([dict(input=blk, limit=len(out), out_cap=len(out) + len(blk) + 64)])
        assert rc == 0 and got == out, (prof, rc)
    print("variant ok:", os.environ.get("LZF_DECOMPRESS_KERNEL", "(default)"), len(cases), "blocks,", len(items), "malformed")


if __name__ == "__main__":
    main()
