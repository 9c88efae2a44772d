"""Analysis builds of the library with one copy-stage phase disabled (LZF_DBG_SKIP bits: 1 batches, 2 match rounds, 4 far matches,
8 literals, 16 flush) -> rust-lz-fear_amd/liblzfear_hip_<hash>.so; prints 'skip path' lines.  Output of such a build is wrong by design."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_lz_fear_amd  # noqa
from rust_lz_fear_amd import build
for k in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16, 30]:
    print(k, build.build_library(defines=[f"LZF_DBG_SKIP={k}"]))
