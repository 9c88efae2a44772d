"""Golden vectors from liblz4 1.9.3 in the regimes where lz-fear claims byte equality with the
C implementation (README.md:5,14-16; tests/output_equivalence.rs compares frames byte-for-byte
with the `lz4` CLI): U16Table raw output == LZ4_compress_default for inputs < 64 KiB, U32Table
raw output == LZ4_compress_fast_continue on a fresh stream.  Known divergences (SURVEY.md
Appendix B, quirk B2) are recorded with "equal_expected": false when the oracle differs.

  python tests/golden/make_liblz4_vectors.py       (build container; needs liblz4.so.1)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import liblz4_ffi as c  # noqa: E402
import oracle_ffi as o  # noqa: E402
import vectors  # noqa: E402


def fp(b):
    return [len(b), "%08x" % o.xxh32(b)]


def main():
    assert c.available()
    out = {"liblz4": c.lib().LZ4_versionString().decode(), "u32": {}, "u16": {}}
    cases = vectors.small_cases() + vectors.medium_cases() + [(f"librs{i}", s) for i, s in enumerate(vectors.LIB_RS_STRINGS)]
    for name, data in cases:
        if len(data) == 0:
            continue   # quirk B4: lz-fear emits nothing, C emits one token byte
        cf = c.compress_fresh_stream(data)
        rc, of = o.compress2(data, kind=o.TABLE_U32)
        assert rc == 0
        out["u32"][name] = {"in": fp(data), "c": fp(cf), "equal_expected": cf == of}
        if len(data) <= 0xFFFF:
            cd = c.compress_default(data)
            rc, od = o.compress2(data, kind=o.TABLE_U16)
            assert rc == 0
            out["u16"][name] = {"in": fp(data), "c": fp(cd), "equal_expected": cd == od}
    json.dump(out, open(os.path.join(HERE, "liblz4_vectors.json"), "w"), indent=1, sort_keys=True)
    for k in ("u32", "u16"):
        ne = [n for n, v in out[k].items() if not v["equal_expected"]]
        print(k, len(out[k]), "cases,", len(ne), "diverge from C:", ne[:20])


if __name__ == "__main__":
    main()
