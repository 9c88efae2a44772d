#!/usr/bin/env python3
"""delz4 — .lz4 frame -> file on the MI355X codec (the reference's examples/delz4.rs)."""
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
    ap.add_argument("--dictionary", help="dictionary file")
    a = ap.parse_args()
    frame = open(a.input, "rb").read()
    d = open(a.dictionary, "rb").read() if a.dictionary else b""
    info = framed.read_header(frame)
    cap = int(info.content_size) + 64 if info.has_content_size else None
    try:
        data = framed.decompress_frame(frame, dictionary=d, cap=cap)
    except framed.FrameError as e:
        print(f"{a.input}: {e}", file=sys.stderr)
        sys.exit(1)
    open(a.output, "wb").write(data)
    print(f"{a.input}: {len(frame)} -> {len(data)} bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
