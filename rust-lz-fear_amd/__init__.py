"""MI355X-native LZ4 raw-block codec, drop-in for lz-fear's `raw` (and `framed`) modules.

The product is the C-ABI shared library built from csrc/ (include/lzfear_hip.h).  This Python
package is plumbing for tests and bench.py: a ctypes binding (`ffi`), the build recipe
(`build`) and mirrors of the reference's `raw` / `framed` interfaces on top of the C ABI.
There is no CPU fallback anywhere in this package.
"""
from . import build  # noqa: F401
