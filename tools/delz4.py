#!/usr/bin/env python3
"""delz4 — .lz4 frame -> file on the MI355X codec (the reference's examples/delz4.rs).

Like the original (examples/delz4.rs:31-38: `loop { let buf = reader.fill_buf()?; if buf.is_empty() { break }; out.write_all(buf)?;
let n = buf.len(); reader.consume(n) }`) it STREAMS: the frame is read from the file block by block through
framed.LZ4FrameReader (fill_buf / consume; independent blocks are read ahead and decoded a batch at a time), and every decoded block
goes straight to the output file — memory in use is `readahead` blocks, not the file."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import framed  # noqa: E402


def decompress_file(path_in, path_out, dictionary=b"", readahead=16):
    n_out = 0
    with open(path_in, "rb") as fin, open(path_out, "wb") as fout:
        reader = framed.LZ4FrameReader(fin, dictionary=dictionary, readahead=readahead)
        while True:
            buf = reader.fill_buf()
            if not buf:
                break
            fout.write(buf)
            n_out += len(buf)
            reader.consume(len(buf))
        n_in = fin.tell()
    return n_in, n_out


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("input")
    ap.add_argument("output")
    ap.add_argument("--dictionary", help="dictionary file")
    ap.add_argument("--readahead", type=int, default=16, help="independent blocks decoded per batch")
    a = ap.parse_args()
    d = open(a.dictionary, "rb").read() if a.dictionary else b""
    try:
        n_in, n_out = decompress_file(a.input, a.output, dictionary=d, readahead=a.readahead)
    except framed.FrameError as e:
        print(f"{a.input}: {e}", file=sys.stderr)
        sys.exit(1)
    print(f"{a.input}: {n_in} -> {n_out} bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
