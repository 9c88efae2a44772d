#!/usr/bin/env python3
"""dolz4 — file -> .lz4 frame on the MI355X codec.  Equivalent of the reference's examples/dolz4.rs
(CompressionSettings::default().content_checksum(true).independent_blocks(true).compress_with_size),
with the settings exposed as flags instead of being edited in the source."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import framed  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("input")
    ap.add_argument("output")
    ap.add_argument("--block-size", type=int, default=4 << 20, choices=[64 << 10, 256 << 10, 1 << 20, 4 << 20])
    ap.add_argument("--linked", action="store_true", help="independent_blocks(false)")
    ap.add_argument("--block-checksums", action="store_true")
    ap.add_argument("--no-content-checksum", action="store_true")
    ap.add_argument("--no-content-size", action="store_true")
    ap.add_argument("--dictionary", help="dictionary file")
    ap.add_argument("--dictionary-id", type=int, default=0)
    a = ap.parse_args()
    s = framed.CompressionSettings().block_size(a.block_size).independent_blocks(not a.linked)
    s.block_checksums(a.block_checksums).content_checksum(not a.no_content_checksum)
    if a.dictionary:
        s.dictionary(a.dictionary_id, open(a.dictionary, "rb").read())
    data = open(a.input, "rb").read()
    frame = s.compress(data) if a.no_content_size else s.compress_with_size(data)
    open(a.output, "wb").write(frame)
    print(f"{a.input}: {len(data)} -> {len(frame)} bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
