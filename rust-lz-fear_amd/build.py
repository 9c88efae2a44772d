"""Build recipe for liblzfear_hip.so (hipcc, gfx950 only).  In-tree output so that the
library travels with the repo snapshot to the GPU box.

Two flavours from the same sources:
  product   liblzfear_hip.so            the kernels the C ABI launches, no environment knobs
  analysis  liblzfear_hip_analysis.so   -DLZF_ANALYSIS: every kernel generation kept for A/B runs and counter
                                        studies (tools/, the variant parity test), selected with LZF_DECOMPRESS_KERNEL
                                        / LZF_COMPRESS_KERNEL / LZF_*_ORDER
Objects are compiled one source per hipcc process, in parallel, into rust-lz-fear_amd/_obj/<flavour>/.
"""
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("LZF_LIB_PATH") or os.path.join(PKG_DIR, "liblzfear_hip.so")   # override: analysis builds only
ANALYSIS_LIB_PATH = os.path.join(PKG_DIR, "liblzfear_hip_analysis.so")

PRODUCT_HIP = ["capi.hip", "lz4_decompress_batched.hip", "lz4_decompress_paired.hip", "lz4_decompress_seg.hip", "lz4_decompress_fed.hip", "lz4_compress.hip",
               "lz4_compress_compact.hip", "lz4_compress_team.hip", "aux_kernels.hip"]
ANALYSIS_HIP = ["analysis/lz4_decompress.hip"]
CXX_SOURCES = ["frame.cpp", "host_staging.cpp"]
HEADERS = ["kernels.h", "lzf_device.h", "lz4_decompress_batch_phase.inc", "lz4_decompress_parse_phase.inc", "lz4_decompress_feed_phase.inc",
           "analysis/capi_analysis.inc",
           "lzf_copy_helpers.h", "lzf_parse_helpers.h", "lzf_compress_common.h", "lzf_simt.h", "lz4_compress_team.inc",
           "host_staging.h",
           os.path.join(ROOT, "include", "lzfear_hip.h"), os.path.join(ROOT, "include", "lzfear_frame.h")]


def _path(d):
    return d if os.path.isabs(d) else os.path.join(CSRC, d)


def hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return "hipcc"


def _sources(analysis):
    return PRODUCT_HIP + (ANALYSIS_HIP if analysis else []) + CXX_SOURCES


def _newest_header():
    return max((os.path.getmtime(_path(h)) for h in HEADERS if os.path.exists(_path(h))), default=0.0)


LAST_BUILD = {}      # out path -> "compiled N of M sources + linked" | "reused (newer than its sources)": what build_library did last


def build_library(force=False, verbose=False, defines=(), out=None, analysis=False):
    """hipcc --offload-arch=gfx950 -> rust-lz-fear_amd/liblzfear_hip.so (or `out`).  LAST_BUILD[out] says whether hipcc ran."""
    defines = list(defines)
    if (analysis or any(d.startswith("LZF_DBG") for d in defines)) and "LZF_ANALYSIS" not in defines:
        defines.append("LZF_ANALYSIS")
    analysis = "LZF_ANALYSIS" in defines
    extra = sorted(d for d in defines if d != "LZF_ANALYSIS")
    if out is None and extra:       # an instrumented build never takes the name (and the up-to-date check) of the plain analysis library
        out = os.path.join(PKG_DIR, "liblzfear_hip_" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:10] + ".so")
    out = out or (ANALYSIS_LIB_PATH if analysis else LIB_PATH)
    srcs_t = max([os.path.getmtime(_path(s)) for s in _sources(analysis)] + [_newest_header()])
    if not force and os.path.exists(out) and os.path.getmtime(out) >= srcs_t:
        LAST_BUILD[out] = "reused (the library is newer than every source and header)"
        if verbose:
            print(f"[build] {os.path.basename(out)}: {LAST_BUILD[out]}", flush=True)
        return out                                  # (also the GPU box's case: the built library travels, the objects do not)
    tag = "product" if not defines else hashlib.sha1(" ".join(sorted(defines)).encode()).hexdigest()[:12]
    objdir = os.path.join(PKG_DIR, "_obj", tag)
    os.makedirs(objdir, exist_ok=True)
    hdr_t = _newest_header()
    base = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include")] + [f"-D{d}" for d in defines]
    jobs = []
    for s in _sources(analysis):
        src = _path(s)
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(s))[0] + ".o")
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t)
        jobs.append((src, obj, stale))
    todo = [(s, o) for s, o, st in jobs if st]

    def compile_one(so):
        cmd = base + ["-c", so[0], "-o", so[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, todo))
    objs = [o for _, o, _ in jobs]
    if todo or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    LAST_BUILD[out] = f"compiled {len(todo)} of {len(jobs)} sources with hipcc --offload-arch=gfx950, linked"
    if verbose:
        print(f"[build] {os.path.basename(out)}: {LAST_BUILD[out]}", flush=True)
    return out


DIST_LIB_PATH = os.path.join(PKG_DIR, "liblzfear_dist.so")


def build_dist_library(force=False, verbose=False):
    """liblzfear_dist.so: the frame reassembly of the block-sharded compressor over RCCL (include/lzfear_dist.h, csrc/dist_gather.hip).
    Its own library so that the codec library keeps depending on libamdhip64 alone; links librccl and liblzfear_hip."""
    core = build_library(force=False, verbose=verbose)
    src = _path("dist_gather.hip")
    deps = [src, os.path.join(ROOT, "include", "lzfear_dist.h"), os.path.join(ROOT, "include", "lzfear_hip.h")]
    out = DIST_LIB_PATH
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in deps):
        LAST_BUILD[out] = "reused (the library is newer than every source and header)"
    else:
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-o", out, src,
               "-L", os.path.dirname(core), "-l:" + os.path.basename(core), "-L", "/opt/rocm/lib", "-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        LAST_BUILD[out] = "compiled dist_gather.hip with hipcc --offload-arch=gfx950, linked against librccl + liblzfear_hip"
    if verbose:
        print(f"[build] {os.path.basename(out)}: {LAST_BUILD[out]}", flush=True)
    return out


def build_analysis_library(force=False, verbose=False):
    return build_library(force=force, verbose=verbose, analysis=True)


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True, analysis="--analysis" in sys.argv))
