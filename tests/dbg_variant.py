"""Quick parity probe of the decompress variant LZF_DECOMPRESS_KERNEL selects (first mismatch / status per case)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import oracle_ffi as o, vectors
import rust_lz_fear_amd
from rust_lz_fear_amd import ffi
cases = vectors.medium_cases()[:3] + vectors.small_cases()[-3:]
for name, d in cases:
    c = o.compress2(d)[1]
    (rc, out), = ffi.decompress_blocks_host([dict(input=c, limit=max(len(d),1), out_cap=len(d)+len(c)+64)])
    a = np.frombuffer(out, np.uint8); b = np.frombuffer(d, np.uint8)
    n = min(len(a), len(b))
    bad = np.nonzero(a[:n] != b[:n])[0]
    print(name, 'len', len(d), 'clen', len(c), 'rc', rc, 'outlen', len(out), 'first bad', (int(bad[0]), len(bad)) if len(bad) else None)
