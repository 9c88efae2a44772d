"""Build recipe for liblzfear_hip.so (hipcc, gfx950 only).  In-tree output so that the
library travels with the repo snapshot to the GPU box."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("LZF_LIB_PATH") or os.path.join(PKG_DIR, "liblzfear_hip.so")   # override: debug builds only

HIP_SOURCES = ["capi.hip", "lz4_decompress.hip", "lz4_decompress_batched.hip", "lz4_decompress_windowed.hip", "lz4_decompress_paired.hip", "lz4_decompress_walk.hip", "lz4_decompress_v4.hip", "lz4_decompress_v5.hip", "lz4_decompress_v6.hip", "lz4_compress.hip", "lz4_compress_compact.hip", "aux_kernels.hip"]
CXX_SOURCES = ["frame.cpp"]
DEPS = HIP_SOURCES + CXX_SOURCES + ["kernels.h", "lzf_device.h", "lz4_decompress_batch_phase.inc", "lz4_decompress_parse_phase.inc", "lz4_decompress_walk_phase.inc", "lz4_decompress_copy2.inc", "lz4_decompress_copy3.inc", "lz4_decompress_gwalk_phase.inc", "lzf_copy_helpers.h", "lzf_parse_helpers.h", "lzf_compress_common.h", os.path.join(ROOT, "include", "lzfear_hip.h"),
                                     os.path.join(ROOT, "include", "lzfear_frame.h")]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for d in DEPS:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return "hipcc"


def build_library(force=False, verbose=False, defines=(), out=None):
    """hipcc --offload-arch=gfx950 -> rust-lz-fear_amd/liblzfear_hip.so"""
    out = out or LIB_PATH
    if not force and not defines and not _stale():
        return out
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "include"), "-o", out] + [f"-D{d}" for d in defines]
    cmd += [os.path.join(CSRC, s) for s in HIP_SOURCES + CXX_SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out
