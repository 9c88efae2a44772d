"""Analysis builds with the section timers of the copy / feed stages (LZF_DBG_PHASE_SEL = 0..8) -> rust-lz-fear_amd/liblzfear_hip_<hash>.so."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_lz_fear_amd  # noqa
from rust_lz_fear_amd import build
for k in [int(x) for x in sys.argv[1:]] or range(9):
    print(k, os.path.basename(build.build_library(defines=[f"LZF_DBG_PHASE_SEL={k}"])))
