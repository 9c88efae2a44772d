#!/usr/bin/env python3
"""dolz4 — file -> .lz4 frame on the MI355X codec.  Equivalent of the reference's examples/dolz4.rs
(`CompressionSettings::default().content_checksum(true).independent_blocks(true).compress_with_size(&mut file_in, &mut file_out)`,
examples/dolz4.rs:10-17), with the settings exposed as flags instead of being edited in the source.

Like the original it STREAMS: the input is read in 16 MiB pieces into the frame writer (lzf_frame_writer_*, the streaming form of
compress_internal, src/framed/compress.rs:160-282) and the frame's bytes go to the output file as the writer hands them out, so
the memory in use is bounded by the piece size and the writer's blocks in flight, not by the file."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import framed  # noqa: E402

PIECE = 16 << 20


def compress_file(settings, path_in, path_out, with_size=True, piece=PIECE):
    """compress_with_size (src/framed/compress.rs:147-157: the length comes from the file's metadata) / compress (:137-140)."""
    size = os.path.getsize(path_in) if with_size else None
    n_in = n_out = 0
    with open(path_in, "rb") as fin, open(path_out, "wb") as fout:
        def sink(b):
            nonlocal n_out
            fout.write(b); n_out += len(b)
        w = settings.writer(sink, content_size=size)
        try:
            while True:
                chunk = fin.read(piece)
                if not chunk:
                    break
                w.write(chunk); n_in += len(chunk)
            w.finish()
        finally:
            w.close()
    return n_in, n_out


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
    ap.add_argument("--piece", type=int, default=PIECE, help="bytes read from the input per write to the frame writer")
    a = ap.parse_args()
    s = framed.CompressionSettings().block_size(a.block_size).independent_blocks(not a.linked)
    s.block_checksums(a.block_checksums).content_checksum(not a.no_content_checksum)
    if a.dictionary:
        s.dictionary(a.dictionary_id, open(a.dictionary, "rb").read())
    n_in, n_out = compress_file(s, a.input, a.output, with_size=not a.no_content_size, piece=a.piece)
    print(f"{a.input}: {n_in} -> {n_out} bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
