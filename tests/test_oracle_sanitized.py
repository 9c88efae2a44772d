"""The reference's memory-safety story is `#![forbid(unsafe_code)]` + cargo-fuzz; the C restatement
gets an ASan/UBSan run over malformed inputs instead.  Runs the sanitized oracle in
a subprocess (libasan must be first in the link order, hence LD_PRELOAD)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.join(os.path.dirname(HERE), "oracle")

SCRIPT = r'''
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, HERE_PATH); sys.path.insert(0, os.path.dirname(HERE_PATH))
import oracle_ffi as o
o._lib = None
o.build = lambda force=False: SO_PATH
import vectors
rng = np.random.default_rng(5)
n = 0
for name, d in vectors.small_cases()[::2] + vectors.medium_cases()[:6]:
    rc, comp = o.compress2(d)
    assert rc == 0
    assert o.decompress_raw(comp, cap=len(d) + 64) == (0, d)
    rc, frame = o.frame_compress(d, o.make_settings(block_size=65536, independent_blocks=bool(n & 1), block_checksums=bool(n & 2)))
    assert o.frame_decompress(frame)[:2] == (0, d)
    for k in range(8):                       # malformed: exact-size buffers so that any overrun trips ASan
        b = bytearray(comp)
        for _ in range(1 + k % 3):
            if b: b[rng.integers(0, len(b))] = rng.integers(0, 256)
        if k & 1 and len(b) > 1: del b[rng.integers(1, len(b)):]
        lim = max(len(d), 1)
        o.decompress_raw(bytes(b), limit=lim, cap=lim + len(b))
        f = bytearray(frame)
        f[rng.integers(0, len(f))] = rng.integers(0, 256)
        o.frame_decompress(bytes(f), cap=1 << 20)
        n += 1
print("sanitized ok", n)
'''


@pytest.mark.timeout(600)
def test_oracle_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    subprocess.check_call(["make", "-C", ORACLE, "-s", "liblzf_oracle_asan.so"])
    so = os.path.join(ORACLE, "liblzf_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", f"HERE_PATH = {HERE!r}\nSO_PATH = {so!r}\n" + SCRIPT], env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitized ok" in r.stdout
