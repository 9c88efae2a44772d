"""Regenerates the reference-derived fixtures.  Runs ONLY in the build container, where the
reference tree is mounted at /root/reference; nothing here travels except the outputs.

  python tests/golden/make_fixtures.py
"""
import collections
import glob
import json
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_ffi as o  # noqa: E402

REF = "/root/reference"


def main():
    # 1. issue-15 data vector (tests/issue-15.rs:5)
    src = open(os.path.join(REF, "tests", "issue-15.rs")).read()
    m = re.search(r"let input = \[(.*?)\];", src, re.S)
    vals = bytes(int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{2})", m.group(1)))
    assert len(vals) == 81248
    open(os.path.join(HERE, "issue15_input.bin"), "wb").write(vals)

    # 2. the one corpus frame that is pure data (random bytes, stored block)
    shutil.copyfile(os.path.join(REF, "fuzz/corpus/decode/uncomp.data.lz4"), os.path.join(HERE, "uncomp.data.lz4"))

    # 3. fingerprints of the valid frames + census of the whole decode corpus
    out = {"valid_frames": {}, "census": {}}
    census = collections.Counter()
    per_file = {}
    for f in sorted(glob.glob(os.path.join(REF, "fuzz/corpus/decode/*"))):
        data = open(f, "rb").read()
        rc, dec, used = o.frame_decompress(data)
        name = o.STATUS_NAMES[rc]
        census[name] += 1
        per_file[os.path.basename(f)] = name
        if rc == 0 and os.path.basename(f).endswith(".lz4"):
            out["valid_frames"][os.path.basename(f)] = {
                "in_len": len(data), "in_xxh32": "%08x" % o.xxh32(data),
                "out_len": len(dec), "out_xxh32": "%08x" % o.xxh32(dec)}
    out["census"] = dict(sorted(census.items()))
    out["per_file"] = per_file
    json.dump(out, open(os.path.join(HERE, "corpus_frames.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out["census"], indent=1), out["valid_frames"])


if __name__ == "__main__":
    main()
