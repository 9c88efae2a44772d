"""Deterministic synthetic inputs for the BASELINE.json configs.

enwik8 / Silesia are not available offline, so every benchmark and parity input is generated
here: splitmix64 streams, seed = 0x5EED0000 + config number.  Every generator is *piecewise*:
a buffer is a list of segments, each segment is generated in independent 1 MiB pieces whose
seed depends only on (config seed, segment index, piece index) — any byte range (e.g. the
4 MiB blocks one GPU owns) can be produced without generating the rest.
"""
import numpy as np

MASK64 = (1 << 64) - 1
PIECE = 1 << 20

# ------------------------------------------------------------------------------------------
# PRNG
# ------------------------------------------------------------------------------------------

def splitmix64(seed, n):
    """n 64-bit outputs of splitmix64 seeded with `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = (np.uint64(seed & MASK64) + idx * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _mix(*vals):
    h = 0x243F6A8885A308D3
    for v in vals:
        h = (h ^ (v & MASK64)) * 0x9E3779B97F4A7C15 & MASK64
        h ^= h >> 29
    return h


def _uniform(seed, n):
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def lcg_bytes(seed, n, zero_mask=0):
    """The formula-defined KAT generator of SURVEY.md Appendix B/C (pure LCG, byte = s >> 56,
    byte forced to 0 when ((s >> 40) & zero_mask) != 0).  Sequential, so kept small."""
    out = np.empty(n, dtype=np.uint8)
    s = seed & MASK64
    a, c = 6364136223846793005, 1442695040888963407
    for i in range(n):
        s = (s * a + c) & MASK64
        b = s >> 56
        if zero_mask and ((s >> 40) & zero_mask):
            b = 0
        out[i] = b
    return out.tobytes()


# ------------------------------------------------------------------------------------------
# generator classes: each returns exactly n bytes (uint8 array) for one piece
# ------------------------------------------------------------------------------------------
_VOCAB_CACHE = {}


def _vocab(seed, nwords=4096, lo=2, hi=12, alphabet=b"etaoinshrdlucmfwypvbgkqjxz"):
    key = (seed, nwords, lo, hi, alphabet)
    if key not in _VOCAB_CACHE:
        r = splitmix64(_mix(seed, 0xB0CAB), nwords * (hi + 2))
        lens = (r[:nwords] % np.uint64(hi - lo + 1)).astype(np.int64) + lo
        # letters skewed towards the head of the alphabet
        u = (r[nwords:nwords * (hi + 1)] >> np.uint64(11)).astype(np.float64) / (1 << 53)
        letters = np.frombuffer(alphabet, dtype=np.uint8)[(u * u * len(alphabet)).astype(np.int64)]
        W = np.zeros((nwords, hi + 1), dtype=np.uint8)
        W[:, :hi] = letters.reshape(nwords, hi)
        W[np.arange(nwords), lens] = 32  # trailing space
        w = 1.0 / np.arange(1, nwords + 1)
        _VOCAB_CACHE[key] = (W, lens + 1, np.cumsum(w) / w.sum())
    return _VOCAB_CACHE[key]


def _emit_tokens(W, tlen, idx, n):
    """Concatenate rows W[idx[k], :tlen[idx[k]]] and cut to n bytes."""
    L = tlen[idx]
    start = np.cumsum(L) - L
    total = int(L.sum())
    tok = np.repeat(np.arange(len(idx)), L)
    within = np.arange(total) - start[tok]
    out = W[idx[tok], within]
    if total < n:  # pad by repeating (callers over-sample, so this is rare)
        out = np.resize(out, n)
    return out[:n]


def gen_text_zipf(seed, n, vocab_seed=1):
    W, tlen, cdf = _vocab(vocab_seed)
    k = n // 4 + 64
    u = _uniform(seed, k)
    idx = np.searchsorted(cdf, u).clip(0, len(cdf) - 1)
    out = _emit_tokens(W, tlen, idx, n).copy()
    # newlines: roughly every 70-90 characters, on a space
    nl = splitmix64(_mix(seed, 7), n // 64 + 1)
    pos = (np.arange(len(nl)) * 64 + (nl % np.uint64(64)).astype(np.int64))
    pos = pos[pos < n]
    sp = pos[out[pos] == 32]
    out[sp] = 10
    return out


def gen_markup(seed, n, tag_bits=1):
    tags = [b"<row id=\"", b"\"><name>", b"</name><value>", b"</value><ts>2026-09-", b"</ts></row>\n",
            b"<item class=\"", b"\" ref=\"#", b"\"/>\n", b"  <text>", b"</text>\n"]
    width = max(len(t) for t in tags)
    T = np.zeros((len(tags), width), dtype=np.uint8)
    tl = np.zeros(len(tags), dtype=np.int64)
    for i, t in enumerate(tags):
        T[i, :len(t)] = np.frombuffer(t, dtype=np.uint8)
        tl[i] = len(t)
    W, wl, cdf = _vocab(2, nwords=1024, lo=3, hi=9)
    k = n // 6 + 64
    r = splitmix64(seed, k)
    is_tag = (r & np.uint64((1 << tag_bits) - 1)) != np.uint64((1 << tag_bits) - 1) if tag_bits > 1 else (r & np.uint64(1)) == 0
    tag_idx = ((r >> np.uint64(8)) % np.uint64(len(tags))).astype(np.int64)
    # tags mostly follow each other cyclically (structure), sometimes random
    cyc = np.arange(k) % len(tags)
    tag_idx = np.where(((r >> np.uint64(20)) % np.uint64(8)) != 0, cyc, tag_idx)
    word_idx = np.searchsorted(cdf, (r >> np.uint64(11)).astype(np.float64) / (1 << 53)).clip(0, 1023)
    width2 = max(width, W.shape[1])
    M = np.zeros((len(tags) + W.shape[0], width2), dtype=np.uint8)
    M[:len(tags), :width] = T
    M[len(tags):, :W.shape[1]] = W
    ml = np.concatenate([tl, wl])
    idx = np.where(is_tag, tag_idx, word_idx + len(tags))
    return _emit_tokens(M, ml, idx, n)


def gen_exe(seed, n, idiom_every=48, zero_every=4096):
    """Executable-like: skewed opcode bytes, short repeated idioms, zero/0xFF runs, address-like
    little-endian words sharing their high bytes."""
    r = splitmix64(seed, n + 8)
    u = (r[:n] >> np.uint64(40)).astype(np.float64) / (1 << 24)
    common = np.frombuffer(bytes([0x00, 0xFF, 0x48, 0x8B, 0x89, 0xE8, 0x0F, 0x24, 0x44, 0x4C, 0x83, 0xC3, 0x90,
                                  0x01, 0x85, 0xC0, 0x74, 0x75, 0x8D, 0x05, 0x10, 0x20, 0x40, 0x08]), dtype=np.uint8)
    pick = (u ** 3 * len(common)).astype(np.int64)
    out = np.where((r[:n] & np.uint64(3)) != 0, common[pick], (r[:n] >> np.uint64(8)).astype(np.uint8)).astype(np.uint8)
    # idioms: copy a 6-24 byte window from 64..4096 bytes back, every ~48 bytes
    k = n // idiom_every
    rr = splitmix64(_mix(seed, 3), k * 3)
    at = (np.arange(k) * idiom_every + (rr[:k] % np.uint64(idiom_every // 2)).astype(np.int64)) + 4200
    ln = (rr[k:2 * k] % np.uint64(19)).astype(np.int64) + 6
    back = (rr[2 * k:] % np.uint64(4032)).astype(np.int64) + 64
    ok = at + 24 < n
    at, ln, back = at[ok], ln[ok], back[ok]
    for j in range(24):
        m = ln > j
        out[at[m] + j] = out[at[m] + j - back[m]]
    # zero runs: ~6% of the bytes in runs of 16..512
    kz = n // zero_every + 1
    rz = splitmix64(_mix(seed, 5), kz * 2)
    zs = (np.arange(kz) * zero_every + (rz[:kz] % np.uint64(zero_every - 512)).astype(np.int64))
    zl = (rz[kz:] % np.uint64(497)).astype(np.int64) + 16
    for s, l in zip(zs.tolist(), zl.tolist()):
        out[s:min(n, s + l)] = 0
    return out


def gen_records(seed, n, reclen=64, cardinality=48, noise_digits=6):
    """Fixed-width DB/record rows: most fields drawn from small value sets, one counter, one noisy field."""
    rows = n // reclen + 1
    r = splitmix64(seed, rows * 4)
    W, wl, _ = _vocab(3, nwords=cardinality, lo=6, hi=12)
    rec = np.full((rows, reclen), 32, dtype=np.uint8)
    a = (r[:rows] % np.uint64(cardinality)).astype(np.int64)
    b = ((r[rows:2 * rows] >> np.uint64(7)) % np.uint64(cardinality)).astype(np.int64)
    rec[:, 0:13] = W[a]
    rec[:, 14:27] = W[b]
    ctr = np.arange(rows, dtype=np.int64) + int(seed % 100000)
    for d in range(8):
        rec[:, 35 - d] = 48 + (ctr // (10 ** d)) % 10
    noise = r[2 * rows:3 * rows]
    for d in range(noise_digits):
        rec[:, 40 + d] = 48 + ((noise >> np.uint64(8 * d)) % np.uint64(10)).astype(np.uint8)
    rec[:, reclen - 1] = 10
    return rec.reshape(-1)[:n]


def gen_walk16(seed, n, noise_bits=6, background=0.0):
    """Smooth 16-bit little-endian samples (medical-image like): random walk + low-bit noise;
    `background` = fraction of samples inside constant (zero) scanline margins."""
    k = n // 2 + 1
    r = splitmix64(seed, k)
    step = ((r & np.uint64(7)).astype(np.int64) - 3)
    base = np.cumsum(step) * 3 + 20000
    nz = ((r >> np.uint64(16)) & np.uint64((1 << noise_bits) - 1)).astype(np.int64)
    v = ((base >> 3 << 3) + (nz >> 3)).astype(np.int64) & 0xFFFF
    if background > 0.0:
        row = 512   # samples per scanline; the first `margin` of each line are background
        margin = int(row * background)
        v = np.where((np.arange(k) % row) < margin, 0, v)
    out = np.empty(k * 2, dtype=np.uint8)
    out[0::2] = v & 0xFF
    out[1::2] = v >> 8
    return out[:n]


def gen_random(seed, n):
    return splitmix64(seed, n // 8 + 1).view(np.uint8)[:n]


def gen_log(seed, n):
    """Timestamped log lines from 64 templates with numeric fields (config 4, ratio ~4-6)."""
    W, wl, cdf = _vocab(4, nwords=256, lo=4, hi=10)
    nt = 64
    tr = splitmix64(0x10C, nt * 8)
    tmpl = [b" ".join(bytes(W[int(tr[t * 8 + j] % np.uint64(256)), :wl[int(tr[t * 8 + j] % np.uint64(256))] - 1])
                      for j in range(3 + t % 5)) for t in range(nt)]
    lines = n // 40 + 8
    r = splitmix64(seed, lines * 2)
    out = bytearray()
    base_ts = 1790000000 + int(seed % 1000000)
    lv = [b"INFO", b"WARN", b"DEBUG", b"ERROR"]
    # Python loop over lines is too slow for GiB-scale; build with numpy via fixed-width layout
    width = 160
    M = np.full((lines, width), 32, dtype=np.uint8)
    ts = base_ts + np.arange(lines, dtype=np.int64) // 7
    for d in range(10):
        M[:, 9 - d] = 48 + (ts // (10 ** d)) % 10
    M[:, 10] = ord(".")
    ms = (r[:lines] % np.uint64(1000)).astype(np.int64)
    for d in range(3):
        M[:, 13 - d] = 48 + (ms // (10 ** d)) % 10
    lvl = ((r[:lines] >> np.uint64(12)) % np.uint64(16)).astype(np.int64)
    lvl = np.where(lvl < 11, 0, np.where(lvl < 13, 1, np.where(lvl < 15, 2, 3)))
    L = np.zeros((4, 6), dtype=np.uint8) + 32
    for i, s in enumerate(lv):
        L[i, :len(s)] = np.frombuffer(s, dtype=np.uint8)
    M[:, 15:21] = L[lvl]
    tid = ((r[:lines] >> np.uint64(20)) % np.uint64(nt)).astype(np.int64)
    tw = max(len(t) for t in tmpl)
    T = np.full((nt, tw), 32, dtype=np.uint8)
    tlen = np.zeros(nt, dtype=np.int64)
    for i, t in enumerate(tmpl):
        T[i, :len(t)] = np.frombuffer(t, dtype=np.uint8)
        tlen[i] = len(t)
    assert 22 + tw + 8 <= width
    M[:, 22:22 + tw] = T[tid]
    # numeric field after the template text
    val = (r[lines:2 * lines] % np.uint64(100000)).astype(np.int64)
    col = 22 + tlen[tid] + 1
    col = np.minimum(col, width - 8)
    rows = np.arange(lines)
    M[rows, col] = ord("=")
    for d in range(5):
        M[rows, col + 5 - d] = 48 + (val // (10 ** d)) % 10
    end = col + 6
    M[rows, end] = 10
    # variable-length lines: keep bytes up to and including the newline
    keep = np.arange(width)[None, :] <= end[:, None]
    flat = M[keep]
    if len(flat) < n:
        flat = np.resize(flat, n)
    return flat[:n]


def gen_repeat256(seed, n):
    motif = splitmix64(seed, 32).view(np.uint8)[:256]
    return np.resize(motif, n)


CLASSES = {
    "text": gen_text_zipf, "markup": gen_markup, "exe": gen_exe, "records": gen_records,
    "walk16": gen_walk16, "random": gen_random, "log": gen_log,
}

# ------------------------------------------------------------------------------------------
# configs
# ------------------------------------------------------------------------------------------
SILESIA_SEGMENTS = [  # (name, bytes — the public Silesia file sizes, class, kwargs)
    ("dickens", 10192446, "text", {}),
    ("mozilla", 51220480, "exe", {"idiom_every": 20, "zero_every": 1536}),
    ("mr", 9970564, "walk16", {"noise_bits": 5, "background": 0.45}),
    ("nci", 33553445, "records", {"reclen": 80, "cardinality": 24, "noise_digits": 2}),
    ("ooffice", 6152192, "exe", {}),
    ("osdb", 10085684, "records", {"reclen": 48, "cardinality": 512}),
    ("reymont", 6627202, "text", {"vocab_seed": 5}),
    ("samba", 21606400, "markup", {}),
    ("sao", 7251944, "random", {}),
    ("webster", 41458703, "text", {"vocab_seed": 9}),
    ("x-ray", 8474240, "walk16", {"noise_bits": 11}),
    ("xml", 5345280, "markup", {"tag_bits": 2}),
]
SILESIA_TOTAL = sum(s[1] for s in SILESIA_SEGMENTS)   # 211 938 580
assert SILESIA_TOTAL == 211938580


def _segment_piece(cfg_seed, seg_idx, piece_idx, cls, kwargs, n):
    return CLASSES[cls](_mix(cfg_seed, seg_idx, piece_idx), n, **kwargs)


def gen_segments(segments, cfg_seed, start, end):
    """Bytes [start, end) of the concatenation of `segments` (name, length, class, kwargs)."""
    out = np.empty(end - start, dtype=np.uint8)
    pos = 0  # absolute offset of the current segment
    w = 0
    for si, (_, slen, cls, kw) in enumerate(segments):
        lo, hi = max(start, pos), min(end, pos + slen)
        if lo < hi:
            p0, p1 = (lo - pos) // PIECE, (hi - pos - 1) // PIECE
            for pi in range(p0, p1 + 1):
                pstart = pos + pi * PIECE
                plen = min(PIECE, slen - pi * PIECE)
                piece = _segment_piece(cfg_seed, si, pi, cls, kw, plen)
                a, b = max(lo, pstart), min(hi, pstart + plen)
                out[w:w + (b - a)] = piece[a - pstart:b - pstart]
                w += b - a
        pos += slen
    assert w == end - start, (w, start, end)
    return out


def silesia_mix(start=0, end=SILESIA_TOTAL, copy=0):
    """Config 2/3 stand-in for the Silesia corpus; `copy` selects the seed of a tiled copy."""
    return gen_segments(SILESIA_SEGMENTS, 0x5EED0002 + (copy << 32), start, end)


def text_zipf_64k():
    """Config 1: one 65 536-byte block of enwik8-like text."""
    return gen_text_zipf(0x5EED0001, 65536)


def log_text(start, end):
    """Config 4: [start, end) of the 8 GiB log stream."""
    total = 8 << 30
    return gen_segments([("log", total, "log", {})], 0x5EED0004, start, end)


def repeat256(n, seed=0x5EED0005):
    """Config 5: a seeded 256-byte motif repeated."""
    return gen_repeat256(seed, n)
