"""Stage-by-stage check of the segmented decompress pipeline (lz4_decompress_seg.hip) against a host parse of the same
blocks: token bit maps after the seam stage, token / output totals after the scan, records after the record stage, the
validity of the dependency levels, and the decoded bytes.  Needs the ANALYSIS library (lzf_debug_seg is not in the product):
    LZF_LIB_PATH=rust-lz-fear_amd/liblzfear_hip_analysis.so python tools/seg_debug.py [--big N] [--small]
ANALYSIS TOOL: uses the oracle as the checker."""
import argparse
import ctypes as C
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import oracle_ffi as o  # noqa: E402
import rust_lz_fear_amd  # noqa: E402,F401
from rust_lz_fear_amd import ffi, device, synth  # noqa: E402

CHUNK, OVERLAP, TILE = 16384, 2048, 2048
STRIDE = CHUNK - OVERLAP
SEGJOB = np.dtype([("eligible", "<u4"), ("failed", "<u4"), ("done", "<u4"), ("nch", "<u4"), ("ntile", "<u4"), ("ntok", "<u4"),
                   ("outb", "<u4"), ("pad", "<u4"), ("rec_off", "<u8"), ("pad2", "<u8")])


def host_parse(c):
    """[(pos, L, M, off, src)] of a valid block (decompress.rs:61-74)."""
    out = []
    p, n = 0, len(c)
    while p < n:
        pos = p
        tok = c[p]; p += 1
        L = tok >> 4
        if L == 15:
            while True:
                b = c[p]; p += 1; L += b
                if b != 255:
                    break
        src = p
        p += L
        if n - p < 2:
            out.append((pos, L, 0, 0, src))
            break
        off = c[p] | (c[p + 1] << 8); p += 2
        M = tok & 15
        if M == 15:
            while True:
                b = c[p]; p += 1; M += b
                if b != 255:
                    break
        out.append((pos, L, M + 4, off, src))
    return out


def run(blocks, min_in, upto, verbose=True):
    """blocks: list of (name, raw bytes).  Returns True when every check passed."""
    dev = torch.device("cuda:0")
    comps = []
    for name, d in blocks:
        rc, cdat = o.compress2(d)
        assert rc == 0, name
        comps.append(cdat)
    n = len(blocks)
    in_off = np.zeros(n, np.uint64); out_off = np.zeros(n, np.uint64)
    ti = to = 0
    for i, ((_, d), cdat) in enumerate(zip(blocks, comps)):
        in_off[i] = ti; ti += (len(cdat) + 255) // 256 * 256 + 3      # odd alignments on purpose
        out_off[i] = to; to += (len(d) + len(cdat) + 64 + 255) // 256 * 256 + 5
    hin = np.zeros(ti + 64, np.uint8)
    for i, cdat in enumerate(comps):
        hin[int(in_off[i]):int(in_off[i]) + len(cdat)] = np.frombuffer(cdat, np.uint8)
    din = torch.from_numpy(hin).to(dev)
    dout = torch.full((to + 64,), 0xEE, dtype=torch.uint8, device=dev)
    jobs = np.zeros(n, device.DJOB)
    jobs["input"] = np.uint64(din.data_ptr()) + in_off
    jobs["input_len"] = [len(cdat) for cdat in comps]
    jobs["out"] = np.uint64(dout.data_ptr()) + out_off
    jobs["out_cap"] = [len(d) + len(cdat) + 64 for (_, d), cdat in zip(blocks, comps)]
    jobs["output_limit"] = [max(len(d), 1) for _, d in blocks]
    djobs = device.to_device(jobs, dev)
    dres = torch.zeros(n * device.RES.itemsize, dtype=torch.uint8, device=dev)
    lib = ffi.lib()
    fn = lib.lzf_debug_seg
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 7 + [C.c_uint64, C.c_void_p]
    geom = np.zeros(4, np.uint32)
    # first call only for the geometry
    ffi.check(fn(djobs.data_ptr(), dres.data_ptr(), n, min_in, 1, None, None, None, None, None, None, None, 0, geom.ctypes.data))
    maxch, maxtile, cw, rec_cap = [int(x) for x in geom]
    st = np.zeros(n, SEGJOB)
    bits = np.zeros((n, maxch, cw), np.uint32)
    xexit = np.zeros((n, maxch), np.uint32); vfrom = np.zeros((n, maxch), np.uint32)
    ttok = np.zeros((n, maxtile), np.uint32); tout = np.zeros((n, maxtile), np.uint32)
    max_recs = min(rec_cap, sum(len(c) // 3 + 128 for c in comps) + 128 * n)
    recs = np.zeros((max_recs, 4), np.uint32)
    torch.cuda.synchronize()
    t0 = time.time()
    ffi.check(fn(djobs.data_ptr(), dres.data_ptr(), n, min_in, upto, st.ctypes.data, bits.ctypes.data, xexit.ctypes.data, vfrom.ctypes.data,
                 ttok.ctypes.data, tout.ctypes.data, recs.ctypes.data, recs.nbytes, geom.ctypes.data))
    if verbose:
        print(f"lzf_debug_seg upto={upto}: {n} jobs, {time.time() - t0:.3f} s (incl. copies)")
    res = device.results_to_host(dres, n)
    hout = dout.cpu().numpy()
    ok_all = True
    for i, ((name, d), cdat) in enumerate(zip(blocks, comps)):
        s = st[i]
        toks = host_parse(cdat)
        msgs = []
        if not s["eligible"]:
            print(f"  [{name}] not eligible (len {len(cdat)})")
            continue
        nch = int(s["nch"])
        exp_nch = 1 if len(cdat) <= CHUNK else 1 + (len(cdat) - CHUNK + STRIDE - 1) // STRIDE
        if nch != exp_nch:
            msgs.append(f"nch {nch} != {exp_nch}")
        if upto >= 3 and not s["failed"]:
            # the true token set against the stitched bit maps
            truth = np.zeros(len(cdat) + 64, np.uint8)
            for t in toks:
                truth[t[0]] = 1
            got = np.zeros(len(cdat) + 64, np.uint8)
            for h in range(nch):
                base = h * STRIDE
                o0 = 0 if h == 0 else base + OVERLAP
                o1 = min(base + CHUNK, len(cdat)) if h < nch - 1 else len(cdat)
                vf = int(vfrom[i, h])
                if vf == 0xFFFFFFFF or o1 <= o0:
                    continue
                row = np.unpackbits(bits[i, h].view(np.uint8), bitorder="little")
                a = max(o0, vf)
                if a < o1:
                    got[a:o1] = row[a - base:o1 - base]
            diff = np.nonzero(truth[:len(cdat)] != got[:len(cdat)])[0]
            if len(diff):
                p = int(diff[0]); h = 0 if p < CHUNK else 1 + (p - CHUNK) // STRIDE
                msgs.append(f"token map differs at {len(diff)} positions, first {p} (truth {truth[p]}, chunk {h}, vfrom {vfrom[i, h]:#x}, x[h-1] {xexit[i, h - 1] if h else 0})")
        # records live in batches of 64 that never span 2 KiB tiles of the input: a tile's tokens, then padding
        slots = []                                   # index into toks, or None (padding)
        k0 = 0
        ntile = (len(cdat) + TILE - 1) // TILE
        for t in range(ntile):
            k1 = k0
            while k1 < len(toks) and toks[k1][0] < (t + 1) * TILE:
                k1 += 1
            slots.extend(range(k0, k1)); slots.extend([None] * ((-(k1 - k0)) % 64))
            k0 = k1
        if upto >= 5 and not s["failed"]:
            if int(s["ntok"]) != len(slots):
                msgs.append(f"padded record count {s['ntok']} != {len(slots)}")
            if int(s["outb"]) != len(d):
                msgs.append(f"outb {s['outb']} != {len(d)}")
        if upto >= 6 and not s["failed"] and int(s["ntok"]) == len(slots):
            r = recs[int(s["rec_off"]):int(s["rec_off"]) + len(slots)]
            rbias = (int(dout.data_ptr()) + int(out_off[i])) & 15
            lo = 0
            exp = np.zeros((len(slots), 3), np.int64)     # mo, M, off of every slot (padding: the end so far, 0, 0)
            for q, k in enumerate(slots):
                if k is None:
                    exp[q] = (lo, 0, 0)
                else:
                    _, L, M, off, _ = toks[k]
                    exp[q] = (lo + L, M, off); lo += L + M
            got = np.stack([r[:, 1].astype(np.int64) - rbias, r[:, 0], r[:, 3] & 0xFFFF], axis=1)
            bad = np.nonzero((got != exp).any(axis=1))[0]
            if len(bad):
                q = int(bad[0])
                msgs.append(f"{len(bad)} records differ, first slot #{q}: got {got[q].tolist()} exp {exp[q].tolist()}")
            else:
                # levels: a match's level must exceed the level of every match of its batch whose destination it reads
                lv = (r[:, 2] >> 16) & 0xFF
                nbad = 0
                for b0 in range(0, len(slots), 64):
                    for q in range(b0, b0 + 64):
                        mo, M, off = (int(x) for x in exp[q])
                        if not M:
                            continue
                        if lv[q] < 1:
                            nbad += 1; continue
                        s0 = mo - off; e0 = s0 + min(M, off)
                        for q2 in range(b0, q):
                            qmo, qM = int(exp[q2][0]), int(exp[q2][1])
                            if qM and qmo < e0 and qmo + qM > s0 and lv[q2] >= lv[q]:
                                nbad += 1
                                if nbad < 4:
                                    msgs.append(f"level order broken: slot {q} (lvl {lv[q]}) reads slot {q2} (lvl {lv[q2]})")
                if nbad:
                    msgs.append(f"{nbad} level violations")
                elif verbose:
                    mx = [int(lv[b0:b0 + 64].max()) for b0 in range(0, len(slots), 64)]
                    print(f"  [{name}] levels ok, mean max level per batch {sum(mx) / max(1, len(mx)):.2f}, {len(slots) - len(toks)} padding slots for {len(toks)} tokens")
        if upto >= 8:
            got = hout[int(out_off[i]):int(out_off[i]) + len(d)].tobytes()
            if s["failed"] or not s["done"]:
                msgs.append(f"pipeline did not finish the job: failed={s['failed']} done={s['done']}")
            elif res[i]["status"] != 0 or int(res[i]["out_len"]) != len(d):
                msgs.append(f"result status {res[i]['status']} out_len {res[i]['out_len']} (expected {len(d)})")
            elif got != d:
                a = np.frombuffer(got, np.uint8); b = np.frombuffer(d, np.uint8)
                w = np.nonzero(a != b)[0]
                msgs.append(f"output differs at {len(w)} bytes, first {int(w[0])}, last {int(w[-1])}")
            tail = hout[int(out_off[i]) + len(d):int(out_off[i]) + len(d) + 5]
            if (tail != 0xEE).any():
                msgs.append("bytes behind the output were written")
            if i and (hout[int(out_off[i]) - 5:int(out_off[i])] != 0xEE).any():
                msgs.append("bytes in front of the output were written")
        if msgs:
            ok_all = False
            print(f"  [{name}] len {len(d)} comp {len(cdat)} toks {len(toks)} FAILED: " + "; ".join(msgs))
        elif verbose:
            extra = ""
            if int(s["pad"]):
                p2 = int(s["pad2"])
                nr = (int(s['pad']) & 0xFFFF) << 4
                extra = f", rounds {nr} ({nr / max(1, (len(toks) + 63) // 64):.2f}/batch), stager waiting {(int(s['pad']) >> 16) << 4} kcycles, resolver kcycles: wait {(p2 & 0xFFFF) << 4} setup {((p2 >> 16) & 0xFFFF) << 4} asm rounds {((p2 >> 32) & 0xFFFF) << 4} slow paths {((p2 >> 48) & 0xFFFF) << 4}"
            print(f"  [{name}] ok (comp {len(cdat)}, {len(toks)} tokens, failed={s['failed']}, kcycles {res[i]['reserved']}{extra})")
    return ok_all


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", type=int, default=6, help="4 MiB Silesia stand-in blocks to check")
    ap.add_argument("--small", action="store_true", help="also the small / medium vectors with min_in = 0")
    ap.add_argument("--upto", type=int, default=8)
    a = ap.parse_args()
    ok = True
    if a.small:
        import vectors
        cases = [(nm, bytes(d)) for nm, d in vectors.small_cases()[::2] + vectors.medium_cases() if len(d) > 0]
        ok &= run(cases, 0, a.upto, verbose=False)
        print("small/medium vectors:", "ok" if ok else "FAILED")
    if a.big:
        sel = [0, 3, 15, 17, 25, 29, 47, 50, 2, 14, 26, 37][:a.big]
        blocks = []
        for b in sel:
            s, e = b * (4 << 20), min((b + 1) * (4 << 20), synth.SILESIA_TOTAL)
            blocks.append((f"silesia[{b}]", synth.silesia_mix(s, e).tobytes()))
        ok &= run(blocks, 65536, a.upto)
    print("SEG DEBUG:", "ALL OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
