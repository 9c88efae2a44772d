"""ctypes binding of tests/emu/libemu_compress_team.so — TEST INFRASTRUCTURE ONLY.

The library is the source of the team compress kernel of the latency class (rust-lz-fear_amd/csrc/lz4_compress_team.inc: searcher /
emitter / feeder wavefronts per block) compiled with g++ against the lock-step wavefront emulator of lzf_simt.h; see
tests/emu/emu_compress_team.cpp.  The product never loads it.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(_HERE, "emu")
CSRC = os.path.join(os.path.dirname(_HERE), "rust-lz-fear_amd", "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")


class CompressJob(C.Structure):
    _fields_ = [("input", C.c_void_p), ("input_len", C.c_uint64), ("cursor", C.c_uint64), ("out", C.c_void_p),
                ("out_cap", C.c_uint64), ("table", C.c_void_p), ("table_kind", C.c_uint32), ("flags", C.c_uint32)]


class JobResult(C.Structure):
    _fields_ = [("out_len", C.c_uint64), ("status", C.c_int32), ("reserved", C.c_uint32)]


_team = None


def build_team(force=False):
    so = os.path.join(EMU_DIR, "libemu_compress_team.so")
    deps = [os.path.join(EMU_DIR, "emu_compress_team.cpp")] + [os.path.join(CSRC, f) for f in
            ("lz4_compress_team.inc", "lzf_simt.h", "lzf_compress_common.h")] + [os.path.join(INCLUDE, "lzfear_hip.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", so, deps[0]])
    return so


def team_lib():
    global _team
    if _team is None:
        L = C.CDLL(build_team())
        L.lzf_emu_compress_team.restype = C.c_int
        L.lzf_emu_compress_team.argtypes = [C.POINTER(CompressJob), C.POINTER(JobResult), C.c_uint32, C.POINTER(C.c_uint32),
                                            C.c_uint32, C.POINTER(C.c_uint64)]
        _team = L
    return _team


def compress_batch(inputs, cursors=None, caps=None, tables=None, perm=None, pad=64, kernel="team", writable=False, alone=1):
    """Runs the emulated team kernel over a batch.  inputs: list of bytes.  Returns [(status, bytes)], lock-step points per wave sum."""
    n = len(inputs)
    jobs = (CompressJob * n)()
    res = (JobResult * n)()
    keep = []
    outs = []
    for i, data in enumerate(inputs):
        data = bytes(data)
        ib = C.create_string_buffer(data, max(len(data), 1))
        cap = caps[i] if caps is not None and caps[i] is not None else len(data) + len(data) // 255 + 64
        ob = C.create_string_buffer(cap + pad)
        ob.raw  # noqa
        C.memset(ob, 0xCD, cap + pad)
        keep.append(ib)
        outs.append((ob, cap))
        jobs[i].input = C.addressof(ib)
        jobs[i].input_len = len(data)
        jobs[i].cursor = cursors[i] if cursors is not None else 0
        jobs[i].out = C.addressof(ob)
        jobs[i].out_cap = cap
        if tables is not None and tables[i] is not None:
            jobs[i].table = C.addressof(tables[i])
            jobs[i].flags = 0 if writable else 1          # LZF_CJOB_TABLE_READONLY unless the table is the caller's to keep
        jobs[i].table_kind = 0
    permarr = None
    if perm is not None:
        permarr = (C.c_uint32 * n)(*perm)
    ns = C.c_uint64(0)
    assert kernel == "team"
    rc = team_lib().lzf_emu_compress_team(jobs, res, n, permarr, alone, C.byref(ns))
    assert rc == 0, rc
    result = []
    for i in range(n):
        ob, cap = outs[i]
        raw = ob.raw
        assert raw[cap:] == b"\xCD" * pad, "job %d wrote beyond its capacity" % i
        result.append((res[i].status, raw[: res[i].out_len]))
    return result, ns.value
