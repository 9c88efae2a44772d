// lz4_decompress_seg.hip — raw::decompress_raw (src/raw/decompress.rs:58-138) for gfx950 with ONE BLOCK DECODED BY MANY
// WAVEFRONTS: the segmented pipeline of kernels.h (plan, parse, seam, tilesum, scan, records, levels, resolve).
//
// Why: a block is a serial chain twice over — the position of a token depends on every token before it
// (decompress.rs:61-71), and a match copies bytes an earlier match produced (:80-138; on text the chain of dependent
// matches is ~1/16 of the sequences long, tools/seg_depth.c).  One workgroup per block therefore needs ~20 ms per 4 MiB
// block whatever the load.  Here everything that is not the second chain is spread over the chip:
//   * the first chain is cut by SPECULATION: a token walk started at an arbitrary byte is on the true chain within 1-2 KB
//     (tools/seq_stats.c), so every 16 KiB chunk is parsed on its own from its first byte and the chains are joined where
//     the true one steps on a token the next chunk's chain has marked;
//   * literals go to their final place in `out` from all tiles at once (they depend on nothing);
//   * the second chain is reduced to what it is: per batch of 64 sequences, one LDS round (read, write) per dependency
//     level, by a wave that does nothing else — the records it consumes carry positions, lengths and levels.
// Error behaviour: the pipeline only finishes jobs that decode cleanly.  Anything else (every DecodeError, a buffer that
// is too small, prefix / existing output, sizes outside the window) is left to the pair kernel, launched last, which
// skips finished jobs — so statuses and partial outputs are exactly the pair kernel's.
#include "lzf_device.h"
#include "kernels.h"
#include "lzf_copy_helpers.h"
#include <type_traits>

namespace lzf {
namespace {

constexpr uint32_t S = kSegRegion;
constexpr uint32_t kCB = kSegChunk + 64u;          // staged bytes of a chunk: the chunk + room for token heads
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kLenClamp = 1u << 26;           // a literal / match length beyond this sends the job to the pair kernel
constexpr uint32_t kTileStage = kSegTile + 128u;   // staged bytes of a tile
constexpr uint32_t kTileTokMax = kSegTile / 3u + 2u;
constexpr uint32_t kFlagCoop = 1u;                 // record flag: the match is moved by the whole wave (longer than 64 bytes)

struct __attribute__((packed, aligned(1))) S8B { uint64_t v; };
struct __attribute__((packed, aligned(1))) S4B { uint32_t v; };
struct __attribute__((packed, aligned(1))) S2B { uint16_t v; };
__device__ __forceinline__ void st8(gu8* p, uint64_t v) { reinterpret_cast<LZF_GLOBAL S8B*>(p)->v = v; }
__device__ __forceinline__ void st4(gu8* p, uint32_t v) { reinterpret_cast<LZF_GLOBAL S4B*>(p)->v = v; }
__device__ __forceinline__ void st2(gu8* p, uint32_t v) { reinterpret_cast<LZF_GLOBAL S2B*>(p)->v = (uint16_t)v; }

// exact per-lane copy of n (1..64) bytes, global -> global, ranges not overlapping: two-ended pieces
__device__ __forceinline__ void copy_small_gg(gu8* d, cgu8* g, uint32_t n) {
    if (n > 32u) {
        const uint64_t a0 = ld8(g), a1 = ld8(g + 8u), a2 = ld8(g + 16u), a3 = ld8(g + 24u);
        const uint64_t b0 = ld8(g + n - 32u), b1 = ld8(g + n - 24u), b2 = ld8(g + n - 16u), b3 = ld8(g + n - 8u);
        st8(d, a0); st8(d + 8u, a1); st8(d + 16u, a2); st8(d + 24u, a3);
        st8(d + n - 32u, b0); st8(d + n - 24u, b1); st8(d + n - 16u, b2); st8(d + n - 8u, b3);
    } else if (n >= 16u) {
        const uint64_t a0 = ld8(g), a1 = ld8(g + 8u), b0 = ld8(g + n - 16u), b1 = ld8(g + n - 8u);
        st8(d, a0); st8(d + 8u, a1); st8(d + n - 16u, b0); st8(d + n - 8u, b1);
    } else if (n >= 8u) {
        const uint64_t a0 = ld8(g), b0 = ld8(g + n - 8u);
        st8(d, a0); st8(d + n - 8u, b0);
    } else if (n >= 4u) {
        const uint32_t a0 = ld4(g), b0 = ld4(g + n - 4u);
        st4(d, a0); st4(d + n - 4u, b0);
    } else if (n >= 2u) {
        const uint32_t a0 = ld2(g), b0 = ld2(g + n - 2u);
        st2(d, a0); st2(d + n - 2u, b0);
    } else if (n == 1u) {
        d[0] = g[0];
    }
}

// per-lane copy of n (65..256) bytes, global -> global, ranges not overlapping: 16-byte pieces, the last one two-ended
__device__ __forceinline__ void copy_medium_gg(gu8* d, cgu8* g, uint32_t n) {
    uint32_t t = 0;
    for (; t + 64u <= n; t += 64u) {
        const u32x4 a0 = ld16(g + t), a1 = ld16(g + t + 16u), a2 = ld16(g + t + 32u), a3 = ld16(g + t + 48u);
        st16(d + t, a0); st16(d + t + 16u, a1); st16(d + t + 32u, a2); st16(d + t + 48u, a3);
    }
    for (; t + 16u <= n; t += 16u) st16(d + t, ld16(g + t));
    if (t < n) st16(d + n - 16u, ld16(g + n - 16u));
}
// cooperative copy of a long run, 4 KiB per step with four loads in flight per lane
__device__ __forceinline__ void wave_copy_long(gu8* __restrict__ dst, cgu8* __restrict__ src, uint32_t n, uint32_t lane) {
    uint32_t i = 0;
    for (; i + 4096u <= n; i += 4096u) {
        const uint32_t o = i + lane * 16u;
        const u32x4 a0 = ld16(src + o), a1 = ld16(src + o + 1024u), a2 = ld16(src + o + 2048u), a3 = ld16(src + o + 3072u);
        st16(dst + o, a0); st16(dst + o + 1024u, a1); st16(dst + o + 2048u, a2); st16(dst + o + 3072u, a3);
    }
    wave_copy(dst + i, src + i, n - i, lane);
}

__device__ __forceinline__ uint32_t seg_nch(uint32_t len) {
    return len <= kSegChunk ? 1u : 1u + (len - kSegChunk + kSegStride - 1u) / kSegStride;
}

// Length of the run of 0xFF bytes that starts at q (stops at len): the body of read_lsic (decompress.rs:30-43) eight bytes
// at a time.  RD8(q) = 8 bytes at q (q + 8 <= len), RDB(q) = one byte.
template <class RD8, class RDB>
__device__ __forceinline__ uint32_t count_ff(uint32_t q, uint32_t len, RD8 rd8, RDB rdb) {
    uint32_t n = 0;
    while (len - q >= 8u) {
        const uint64_t w = rd8(q);
        if (w != ~0ull) return n + ((uint32_t)__builtin_ctzll(~w) >> 3);
        n += 8u; q += 8u;
    }
    while (q < len && rdb(q) == 255u) { ++n; ++q; }
    return n;
}
// LSIC value that continues a nibble of 15 at q: false when the input ends before the terminating byte (UnexpectedEnd).
// v = 15 + 255 * run + last byte, clamped; q moves behind the last byte.
template <class RD8, class RDB>
__device__ __forceinline__ bool read_lsic_tail(uint32_t& q, uint32_t len, uint32_t& v, RD8 rd8, RDB rdb) {
    if (q >= len) return false;
    const uint32_t n = count_ff(q, len, rd8, rdb);
    if (q + n >= len) { q = len; return false; }
    const uint64_t vv = 15ull + 255ull * n + rdb(q + n);
    v = vv > kMaxPosB ? kMaxPosB : (uint32_t)vv;
    q += n + 1u;
    return true;
}
// One token at p (p < len), general form: position of the next token; false on UnexpectedEnd.
// decompress.rs:61-71 without the copies.
template <class RD8, class RDB>
__device__ __forceinline__ bool token_next_gen(uint32_t len, uint32_t p, uint32_t& next, RD8 rd8, RDB rdb) {
    const uint32_t tok = rdb(p);
    uint32_t q = p + 1u;
    uint32_t L = tok >> 4;
    if (L == 15u) { if (!read_lsic_tail(q, len, L, rd8, rdb)) return false; }
    if (len - q < L) return false;                    // :67 read_exact
    q += L;
    if (len - q < 2u) { next = len; return true; }    // :70 read_u16 fails: last literals
    q += 2u;
    if ((tok & 15u) == 15u) { uint32_t M; if (!read_lsic_tail(q, len, M, rd8, rdb)) return false; }
    next = q;
    return true;
}
__device__ __forceinline__ bool token_next_glb(cgu8* in, uint32_t len, uint32_t p, uint32_t& next) {
    return token_next_gen(len, p, next, [&](uint32_t q) -> uint64_t { return ld8(in + q); }, [&](uint32_t q) -> uint32_t { return (uint32_t)in[q]; });
}
__device__ __forceinline__ uint64_t lds_ld64u(uint32_t a) {
    uint64_t v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); return v;
}

}  // namespace

// =====================================================================================================================
// plan
// =====================================================================================================================
__global__ __launch_bounds__(256) void lzf_seg_plan_kernel(seg_ctx c) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0u) *c.rec_top = 0ull;
    if (j >= c.n_jobs) return;
    const lzf_decompress_job job = c.jobs[j];
    seg_job s;
    s.failed = 0; s.done = 0; s.ntok = 0; s.outb = 0; s.pad = 0; s.rec_off = 0; s.pad2 = 0;
    s.eligible = (job.prefix_len == 0 && job.out_existing_len == 0 && job.input_len >= c.min_in && job.input_len <= c.max_in &&
                  job.input != nullptr && job.out != nullptr) ? 1u : 0u;
    const uint32_t len = s.eligible ? (uint32_t)job.input_len : 0u;
    s.nch = s.eligible ? seg_nch(len) : 0u;
    s.ntile = (len + kSegTile - 1u) / kSegTile;
    if (s.nch > c.maxch || s.ntile > c.maxtile) { s.eligible = 0; s.nch = 0; s.ntile = 0; }
    c.st[j] = s;
}

// =====================================================================================================================
// parse: the chain of tokens that starts at a chunk's first byte, as a bit map
// =====================================================================================================================
// A chunk is 64 regions of 256 bytes, one per lane, staged in LDS.
//   pass 0   every lane walks from its region start and marks the tokens it visits in row A, keeps its exit;
//   pass k   entry[i] = max(exit[0..i-1]) (lane 0: the chunk's first byte); a lane whose entry changed walks again from it,
//            marking row B, until it steps on a token marked in row A — from there on it IS the pass-0 chain — or leaves
//            the region; repeated until no entry changes (2-3 passes).
//   result   row = (A from the merge position on) | B; exit of the chunk = max of the exits.
// A plain hop is ONE unaligned 4-byte LDS read from position - 1: [previous token's match-length extension byte, token,
// first literal-length extension byte, ...].  Tokens a plain hop cannot express (0xFF length bytes, the last 24 bytes of
// the input) go through the general routine.
__global__ __launch_bounds__(64) void lzf_seg_parse_kernel(seg_ctx c) {
    __shared__ __attribute__((aligned(16))) uint8_t cbufs[16u + kCB + 16u];
    __shared__ __attribute__((aligned(16))) uint32_t rowsA[64u * 8u], rowsB[64u * 8u];
    const uint32_t j = blockIdx.y;
    const seg_job sj = c.st[j];
    if (!sj.eligible) return;
    const uint32_t lane = threadIdx.x & 63u;
    const lzf_decompress_job job = c.jobs[j];
    cgu8* __restrict__ in = as_global(job.input);
    const uint32_t len = (uint32_t)job.input_len;
    uint8_t* const cbuf = cbufs + 16u;
    const uint32_t cbuf_a = lds_addr(cbuf);
    // word w of a lane's row lives at [w * 64 + lane]: the 64 lanes of an access hit 64 different banks
#define ROWA(w) rowsA[(w) * 64u + lane]
#define ROWB(w) rowsB[(w) * 64u + lane]

    for (uint32_t h = blockIdx.x; h < sj.nch; h += gridDim.x) {
        const uint32_t cstart = h * kSegStride;
        __syncthreads();                         // (one wave: orders the LDS re-use between iterations)
        // ---- stage in[cstart, cstart + kCB) (zeros beyond the input)
        {
            const uint32_t avail = len - cstart < kCB ? len - cstart : kCB;
            cgu8* g = in + cstart;
#pragma unroll 1
            for (uint32_t b4 = 0; b4 < kCB; b4 += 4u * 1024u) {
                u32x4 v[4];
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    const uint32_t i = b4 + k * 1024u + lane * 16u;
                    v[k] = u32x4{0, 0, 0, 0};
                    if (i + 16u <= avail) v[k] = ld16(g + i);
                }
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    const uint32_t i = b4 + k * 1024u + lane * 16u;
                    if (i < kCB) *reinterpret_cast<u32x4*>(&cbuf[i]) = v[k];
                }
            }
            { const uint32_t t0 = avail & ~15u; if (t0 < kCB && lane < (avail & 15u)) cbuf[t0 + lane] = g[t0 + lane]; }   // the ragged end of the input
            if (lane < 4u) reinterpret_cast<uint32_t*>(cbufs)[lane] = 0u;      // the byte in front of the chunk is never an extension byte we trust
        }
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) { ROWA(i) = 0u; ROWB(i) = 0u; }
        const uint32_t room = len - cstart;                                      // bytes from the chunk start to the end of the input
        const uint32_t fe = room > 24u ? (room - 24u < kCB ? room - 24u : kCB) : 0u;   // a plain hop lands below fe (chunk-relative)
        const uint32_t rb0 = lane * S, end_r = rb0 + S;

        auto rdb = [&](uint32_t q) -> uint32_t { const uint32_t r_ = q - cstart; if (r_ < kCB) return (uint32_t)cbuf[r_]; return (uint32_t)in[q]; };
        auto rd8 = [&](uint32_t q) -> uint64_t { const uint32_t r_ = q - cstart; if (r_ + 8u <= kCB) return lds_ld64u(cbuf_a + r_); return ld8(in + q); };
        auto token_next = [&](uint32_t p, uint32_t& next) -> bool { return token_next_gen(len, p, next, rd8, rdb); };
        // MODE 0: mark in row A.  MODE 1: stop on a token marked in row A (merged), mark in row B.
        // Walks from r (chunk-relative) to the first token at or beyond end_r; returns where it stopped.
        auto walk = [&](uint32_t r, bool go, bool& merged, auto MODE) -> uint32_t {
            constexpr int mode = decltype(MODE)::value;
            const uint32_t stop = go ? (end_r < fe ? end_r : fe) : 0u;
            for (;;) {
                uint32_t mxp = 0, rprev = r;
                bool act = go && !merged && r < stop;
                while (__any(act)) {
                    uint32_t lo;
                    asm volatile("ds_read_b32 %0, %1" : "=v"(lo) : "v"(cbuf_a + (act ? r : 0u) - 1u) : "memory");
                    uint32_t mk = 0;
                    const uint32_t bi = (act ? r : rb0) - rb0;
                    if (mode == 1) mk = *reinterpret_cast<const volatile uint32_t*>(&ROWA((bi >> 5) & 7u)) >> (bi & 31u);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const uint32_t e0 = lo & 255u;
                    if (act && mxp != 0u && e0 == 255u) { r = rprev; act = false; }   // the previous token's match length goes on: not plain
                    if (act && r >= stop) act = false;
                    if (mode == 1) { if (act && (mk & 1u)) { merged = true; act = false; } }
                    if (act) {
                        const uint32_t L0 = (lo >> 12) & 15u, M0 = (lo >> 8) & 15u, b1 = (lo >> 16) & 255u;
                        const uint32_t isx = L0 == 15u ? 1u : 0u;
                        const uint32_t Lt = L0 + (isx ? b1 : 0u);
                        const uint32_t mx = M0 == 15u ? 1u : 0u;
                        const uint32_t rn = r + 3u + isx + Lt + mx;
                        const bool plain = !(isx && b1 == 255u) && rn < fe;
                        if (plain) {
                            if (mode == 0) ROWA((bi >> 5) & 7u) |= 1u << (bi & 31u); else ROWB((bi >> 5) & 7u) |= 1u << (bi & 31u);
                            rprev = r; r = rn; mxp = mx;
                        } else {
                            act = false;                                  // parked: the general routine takes this token
                        }
                    }
                }
                // the general routine serves parked lanes and lanes near the end of the input
                const uint32_t pa = cstart + r;
                const bool slow = go && !merged && r < end_r && pa < len;
                if (!__any(slow)) break;
                if (slow) {
                    const uint32_t bi = r - rb0;
                    if (mode == 1 && ((ROWA((bi >> 5) & 7u) >> (bi & 31u)) & 1u)) merged = true;
                    else {
                        if (mode == 0) ROWA((bi >> 5) & 7u) |= 1u << (bi & 31u); else ROWB((bi >> 5) & 7u) |= 1u << (bi & 31u);
                        uint32_t nx;
                        if (!token_next(pa, nx)) nx = len;                // (an error on the true chain is found again by the tile stages)
                        r = nx - cstart;
                    }
                }
            }
            return r;
        };
        using M0T = std::integral_constant<int, 0>; using M1T = std::integral_constant<int, 1>;
        const bool in_input = cstart + rb0 < len;
        bool mdummy = false;
        const uint32_t x0 = walk(rb0, in_input, mdummy, M0T{});
        uint32_t X = in_input ? x0 : 0u, walked = rb0, mpos = rb0;
        for (uint32_t pass = 0; pass < 80u; ++pass) {
            const uint32_t entry = wave_prev(wave_scan_max(X), 0u);
            const bool redo = in_input && entry != walked && lane != 0u;
            if (!__any(redo)) break;
            const bool inreg = redo && entry < end_r && cstart + entry < len;
            if (redo) {
                walked = entry;
#pragma unroll
                for (uint32_t i = 0; i < 8u; ++i) ROWB(i) = 0u;
            }
            bool mg = false;
            const uint32_t x1 = walk(inreg ? entry : end_r, inreg, mg, M1T{});
            if (redo) {
                if (!inreg) { X = entry; mpos = end_r; }
                else if (mg) { X = x0; mpos = x1; }
                else { X = x1; mpos = end_r; }
            }
        }
        // ---- the rows and the exit
        {
            const uint32_t mb = mpos - rb0;                    // 0..256: row A is valid from this bit on
            u32x4 o0, o1;
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) {
                const uint32_t wb = mb >> 5;
                const uint32_t keep = i < wb ? 0u : i == wb ? ~((1u << (mb & 31u)) - 1u) : ~0u;
                const uint32_t v = in_input ? ((ROWA(i) & keep) | ROWB(i)) : 0u;
                if (i < 4u) o0[i] = v; else o1[i - 4u] = v;
            }
            LZF_GLOBAL u32x4* dst = (LZF_GLOBAL u32x4*)(c.bits + ((size_t)j * c.maxch + h) * kSegChunkWords + lane * 8u);
            dst[0] = o0; dst[1] = o1;
            const uint32_t xm = __builtin_amdgcn_readlane(wave_scan_max(X), 63);
            uint64_t xe = (uint64_t)cstart + xm;
            if (xe > len) xe = len;
            if (lane == 0u) c.xexit[(size_t)j * c.maxch + h] = (uint32_t)xe;
        }
    }
}

#undef ROWA
#undef ROWB

// =====================================================================================================================
// seam: from which position on is a chunk's chain the true one
// =====================================================================================================================
__global__ __launch_bounds__(64) void lzf_seg_seam_kernel(seg_ctx c) {
    __shared__ uint16_t patch[kSegStride / 3u + 8u];
    constexpr uint32_t kWin = 2048;                                   // compressed bytes a staged window covers
    __shared__ __attribute__((aligned(16))) uint8_t wbytes[kWin + 128u];
    __shared__ uint32_t wbits[kWin / 32u + 2u];
    const uint32_t j = blockIdx.x;
    if (j >= c.n_jobs) return;
    const seg_job sj = c.st[j];
    if (!sj.eligible) return;
    const uint32_t lane = threadIdx.x & 63u;
    const lzf_decompress_job job = c.jobs[j];
    cgu8* __restrict__ in = as_global(job.input);
    const uint32_t len = (uint32_t)job.input_len;
    LZF_GLOBAL uint32_t* const X = (LZF_GLOBAL uint32_t*)c.xexit + (size_t)j * c.maxch;
    LZF_GLOBAL uint32_t* const VF = (LZF_GLOBAL uint32_t*)c.vfrom + (size_t)j * c.maxch;
    LZF_GLOBAL uint32_t* const B = (LZF_GLOBAL uint32_t*)c.bits + (size_t)j * c.maxch * kSegChunkWords;
    auto bit = [&](uint32_t h, uint32_t pos) -> bool {          // pos inside chunk h's range
        const uint32_t r = pos - h * kSegStride;
        return (B[(size_t)h * kSegChunkWords + (r >> 5)] >> (r & 31u)) & 1u;
    };
    if (lane == 0u) VF[0] = 0u;
    bool carry_valid = false; uint32_t carry_e = 0;
    bool failed = false;
    for (uint32_t h0 = 1; h0 < sj.nch && !failed; h0 += 64u) {
        const uint32_t gn = sj.nch - h0 < 64u ? sj.nch - h0 : 64u;
        const uint32_t h = h0 + lane;
        const bool act = lane < gn;
        const uint32_t e = act ? X[h - 1u] : 0u;
        const uint32_t endh = h * kSegStride + kSegChunk;
        // 0: the true chain (if it enters at e) is on this chunk's chain from e on; 1: it jumps over the chunk; 2: neither
        uint32_t code = 0;
        if (act) code = (e >= len || e >= endh) ? 1u : (bit(h, e) ? 0u : 2u);
        uint32_t k = 0;
        while (k < gn && !failed) {
            uint32_t e_true;
            if (!carry_valid) {
                const uint32_t kk0 = first_lane(__ballot(act && lane >= k && code != 0u));
                const uint32_t kk = kk0 < gn ? kk0 : gn;
                if (act && lane >= k && lane < kk) VF[h] = e;
                k = kk;
                if (k >= gn) break;
                e_true = __builtin_amdgcn_readlane(e, k);
            } else e_true = carry_e;
            // ---- chunk hk entered at e_true
            const uint32_t hk = h0 + k;
            const uint32_t base = hk * kSegStride, endk = base + kSegChunk, ostart = base + kSegOverlap;
            if (e_true >= len || e_true >= endk) {
                if (lane == 0u) VF[hk] = kNone;
                carry_valid = true; carry_e = e_true;
            } else if (bit(hk, e_true)) {
                if (lane == 0u) VF[hk] = e_true;
                carry_valid = false;
            } else {
                // walk the true chain until it steps on a marked token of this chunk (or leaves the chunk)
                // (windows of 2 KiB of the input and of the chunk's marks are staged in LDS, so a hop costs LDS reads)
                uint32_t p = e_true, np = 0, mg = 0, err = 0;
                while (p < endk && p < len && !mg && !err) {
                    const uint32_t ws = p;
                    __syncthreads();
                    {
                        const uint32_t avail = len - ws < kWin + 128u ? len - ws : kWin + 128u;
                        for (uint32_t i = lane * 16u; i < kWin + 128u; i += 1024u) {
                            u32x4 v = u32x4{0, 0, 0, 0};
                            if (i + 16u <= avail) v = ld16(in + ws + i);
                            else if (i < avail) { for (uint32_t b = 0; i + b < avail; ++b) v[(b >> 2) & 3u] |= (uint32_t)in[ws + i + b] << ((b & 3u) * 8u); }
                            *reinterpret_cast<u32x4*>(&wbytes[i]) = v;
                        }
                        const uint32_t w0 = (ws - base) >> 5;                       // first word of the marks the window needs
                        for (uint32_t i = lane; i < kWin / 32u + 2u; i += 64u)
                            wbits[i] = w0 + i < kSegChunkWords ? B[(size_t)hk * kSegChunkWords + w0 + i] : 0u;
                    }
                    __syncthreads();
                    if (lane == 0u) {
                        const uint32_t wb_a = lds_addr(wbytes);
                        auto rdb = [&](uint32_t q) -> uint32_t { const uint32_t r_ = q - ws; if (r_ < kWin + 128u) return (uint32_t)wbytes[r_]; return (uint32_t)in[q]; };
                        auto rd8 = [&](uint32_t q) -> uint64_t { const uint32_t r_ = q - ws; if (r_ + 8u <= kWin + 128u) return lds_ld64u(wb_a + r_); return ld8(in + q); };
                        const uint32_t bit0 = ((ws - base) >> 5) << 5;             // chunk-relative position of bit 0 of wbits[0]
                        while (p < endk && p < len && p - ws < kWin) {
                            const uint32_t rbit = p - base - bit0;
                            if (p != e_true && ((wbits[rbit >> 5] >> (rbit & 31u)) & 1u)) { mg = 1; break; }
                            patch[np++] = (uint16_t)(p - ostart);
                            uint32_t nx;
                            if (!token_next_gen(len, p, nx, rd8, rdb)) { err = 1; break; }
                            p = nx;
                        }
                    }
                    p = __builtin_amdgcn_readfirstlane(p); np = __builtin_amdgcn_readfirstlane(np);
                    mg = __builtin_amdgcn_readfirstlane(mg); err = __builtin_amdgcn_readfirstlane(err);
                }
                if (err) { failed = true; break; }
                uint32_t m = mg ? p : endk;
                // clear this chunk's marks in [ostart, m), then set the walked tokens
                {
                    const uint32_t w0 = kSegOverlap / 32u, mr = m - base;      // mr in (kSegOverlap, kSegChunk]
                    for (uint32_t w = w0 + lane; w * 32u < mr; w += 64u) {
                        LZF_GLOBAL uint32_t* wp = &B[(size_t)hk * kSegChunkWords + w];
                        if (w * 32u + 32u <= mr) *wp = 0u;
                        else *wp &= ~((1u << (mr & 31u)) - 1u);
                    }
                    __threadfence();
                    for (uint32_t i = lane; i < np; i += 64u) {
                        const uint32_t r = (uint32_t)patch[i] + kSegOverlap;
                        atomicOr((uint32_t*)&B[(size_t)hk * kSegChunkWords + (r >> 5)], 1u << (r & 31u));
                    }
                    __threadfence();
                }
                if (lane == 0u) VF[hk] = ostart;
                if (mg) carry_valid = false; else { carry_valid = true; carry_e = p; }
            }
            ++k;
        }
    }
    if (failed && lane == 0u) c.st[j].failed = 1u;
}

// =====================================================================================================================
// tiles: enumerate the true tokens of 2 KiB of compressed bytes and decode them
// =====================================================================================================================
namespace {
struct TileCtx {
    uint32_t tstart;     // first byte of the tile
    uint32_t stage_a;    // LDS address of the staged bytes
    uint32_t len;
    cgu8* in;
};
__device__ __forceinline__ uint32_t tile_rdb(const TileCtx& t, uint32_t q) {
    const uint32_t r = q - t.tstart;
    if (r < kTileStage) return lds_ld8(t.stage_a + r);
    return (uint32_t)t.in[q];
}
// The tokens of tile t of job j: positions (tile-relative) into list[], count returned.  Also stages the tile's bytes.
__device__ __forceinline__ uint32_t tile_enumerate(const seg_ctx& c, uint32_t j, uint32_t t, uint32_t lane, uint32_t len, cgu8* in,
                                                   uint8_t* stage, uint16_t* list) {
    const uint32_t tstart = t * kSegTile;
    // stage [tstart, tstart + kTileStage)
    {
        const uint32_t avail = len - tstart < kTileStage ? len - tstart : kTileStage;
        cgu8* g = in + tstart;
        for (uint32_t i = lane * 16u; i < kTileStage; i += 1024u) {
            u32x4 v = u32x4{0, 0, 0, 0};
            if (i + 16u <= avail) v = ld16(g + i);
            else if (i < avail) { for (uint32_t b = 0; i + b < avail; ++b) v[(b >> 2) & 3u] |= (uint32_t)g[i + b] << ((b & 3u) * 8u); }
            *reinterpret_cast<u32x4*>(&stage[i]) = v;
        }
    }
    const uint32_t h = t < kSegChunk / kSegTile ? 0u : 1u + (t - kSegChunk / kSegTile) / (kSegStride / kSegTile);
    const uint32_t widx = t * (kSegTile / 32u) - h * (kSegStride / 32u) + lane;
    uint32_t w = c.bits[((size_t)j * c.maxch + h) * kSegChunkWords + widx];
    const uint32_t vf = c.vfrom[(size_t)j * c.maxch + h];
    const uint32_t wpos = tstart + lane * 32u;                 // position of bit 0
    if (vf == kNone || vf >= wpos + 32u) w = 0u;
    else if (vf > wpos) w &= ~((1u << (vf - wpos)) - 1u);
    if (wpos >= len) w = 0u;
    else if (len - wpos < 32u) w &= (1u << (len - wpos)) - 1u;
    const uint32_t cnt = (uint32_t)__popc(w);
    const uint32_t incl = wave_scan_add(cnt);
    uint32_t k = incl - cnt;
    while (w) { const uint32_t b = (uint32_t)__builtin_ctz(w); list[k++] = (uint16_t)(lane * 32u + b); w &= w - 1u; }
    __syncthreads();
    return __builtin_amdgcn_readlane(incl, 63);
}
struct Tok { uint32_t L, M, off, src; bool err; };
// decompress.rs:61-74 for the token at p (p < len): lengths, offset, where its literals are.  M = 0: the last sequence.
__device__ __forceinline__ Tok tile_decode(const TileCtx& t, uint32_t p) {
    Tok k; k.err = false; k.M = 0; k.off = 0;
    const uint32_t len = t.len;
    uint32_t w;
    if (p - t.tstart + 4u <= kTileStage) { asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(t.stage_a + (p - t.tstart)) : "memory"); }
    else { w = 0; for (uint32_t i = 0; i < 4u && p + i < len; ++i) w |= tile_rdb(t, p + i) << (8u * i); }
    const uint32_t tok = w & 255u;
    uint32_t q = p + 1u;
    uint32_t L = tok >> 4;
    auto rd8 = [&](uint32_t x) -> uint64_t { const uint32_t r = x - t.tstart; if (r + 8u <= kTileStage) return lds_ld64u(t.stage_a + r); return ld8(t.in + x); };
    auto rdb = [&](uint32_t x) -> uint32_t { return tile_rdb(t, x); };
    if (L == 15u) {
        if (((w >> 8) & 255u) != 255u && q < len) { L += (w >> 8) & 255u; ++q; }
        else if (!read_lsic_tail(q, len, L, rd8, rdb) || L > kLenClamp) k.err = true;
    }
    k.L = L; k.src = q;
    if (k.err) return k;
    if (len - q < L) { k.err = true; return k; }           // :67 read_exact
    q += L;
    if (len - q < 2u) return k;                            // :70 read_u16 fails: last literals, no match
    k.off = tile_rdb(t, q) | (tile_rdb(t, q + 1u) << 8);
    q += 2u;
    uint32_t M = tok & 15u;
    if (M == 15u) { if (!read_lsic_tail(q, len, M, rd8, rdb) || M > kLenClamp) k.err = true; }
    k.M = M + 4u;
    return k;
}
}  // namespace

__global__ __launch_bounds__(64) void lzf_seg_tilesum_kernel(seg_ctx c) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[kTileStage];
    __shared__ uint16_t list[kTileTokMax + 64u];
    const uint32_t j = blockIdx.y;
    const seg_job sj = c.st[j];
    if (!sj.eligible || sj.failed) return;
    const uint32_t lane = threadIdx.x & 63u;
    const lzf_decompress_job job = c.jobs[j];
    cgu8* __restrict__ in = as_global(job.input);
    const uint32_t len = (uint32_t)job.input_len;
    for (uint32_t t = blockIdx.x; t < sj.ntile; t += gridDim.x) {
        __syncthreads();
        const uint32_t n = tile_enumerate(c, j, t, lane, len, in, stage, list);
        TileCtx tc{t * kSegTile, lds_addr(stage), len, in};
        uint32_t sum = 0; bool err = false;
        for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
            if (i0 + lane < n) {
                const Tok k = tile_decode(tc, tc.tstart + list[i0 + lane]);
                err = err || k.err;
                sum += k.L + k.M;                          // (each <= 2^26 + 4, at most 11 per lane)
            }
        }
        const uint64_t tot = (uint64_t)__builtin_amdgcn_readlane(wave_scan_add(sum & 0xFFFFu), 63) +
                             ((uint64_t)__builtin_amdgcn_readlane(wave_scan_add(sum >> 16), 63) << 16);
        if (tot > 0x7FFFFFFFull) err = true;
        if (lane == 0u) { c.tile_tok[(size_t)j * c.maxtile + t] = n; c.tile_out[(size_t)j * c.maxtile + t] = (uint32_t)tot; }
        if (__ballot(err) && lane == 0u) c.st[j].failed = 1u;
    }
}

__global__ __launch_bounds__(64) void lzf_seg_scan_kernel(seg_ctx c) {
    const uint32_t j = blockIdx.x;
    if (j >= c.n_jobs) return;
    const seg_job sj = c.st[j];
    if (!sj.eligible || sj.failed) return;
    const uint32_t lane = threadIdx.x & 63u;
    const lzf_decompress_job job = c.jobs[j];
    LZF_GLOBAL uint32_t* const TT = (LZF_GLOBAL uint32_t*)c.tile_tok + (size_t)j * c.maxtile;
    LZF_GLOBAL uint32_t* const TO = (LZF_GLOBAL uint32_t*)c.tile_out + (size_t)j * c.maxtile;
    uint64_t ctok = 0, cout = 0;
    for (uint32_t t0 = 0; t0 < sj.ntile; t0 += 64u) {
        const uint32_t t = t0 + lane;
        const uint32_t a = t < sj.ntile ? TT[t] : 0u, b = t < sj.ntile ? TO[t] : 0u;
        // 64-bit running sums (a job whose output does not fit 31 bits is not ours)
        const uint32_t ia = wave_scan_add(a);
        uint32_t blo = b & 0xFFFFu, bhi = b >> 16;
        const uint32_t ilo = wave_scan_add(blo), ihi = wave_scan_add(bhi);
        const uint64_t ib = (uint64_t)ilo + ((uint64_t)ihi << 16);
        const uint64_t eo = cout + ib - b;
        if (t < sj.ntile) { TT[t] = (uint32_t)(ctok + ia - a); TO[t] = eo > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)eo; }
        ctok += __builtin_amdgcn_readlane(ia, 63);
        cout += (uint64_t)__builtin_amdgcn_readlane(ilo, 63) + ((uint64_t)__builtin_amdgcn_readlane(ihi, 63) << 16);
    }
    if (lane == 0u) {
        const uint64_t cap = job.out_cap > kMaxPosB ? kMaxPosB : job.out_cap;
        bool ok = cout <= cap && ctok < 0x7FFFFFFFull;
        uint64_t off = 0;
        if (ok) {
            const uint64_t need = (ctok + 63ull) / 64ull * 64ull + 64ull;
            off = atomicAdd(c.rec_top, (unsigned long long)need);
            if (off + need > c.rec_cap) ok = false;
        }
        if (!ok) c.st[j].failed = 1u;
        else { c.st[j].ntok = (uint32_t)ctok; c.st[j].outb = (uint32_t)cout; c.st[j].rec_off = off; }
    }
}

// records + literals
__global__ __launch_bounds__(64) void lzf_seg_records_kernel(seg_ctx c) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[kTileStage];
    __shared__ uint16_t list[kTileTokMax + 64u];
    const uint32_t j = blockIdx.y;
    const seg_job sj = c.st[j];
    if (!sj.eligible || sj.failed) return;
    const uint32_t lane = threadIdx.x & 63u;
    const lzf_decompress_job job = c.jobs[j];
    cgu8* __restrict__ in = as_global(job.input);
    gu8* out = as_global(job.out);
    const uint32_t len = (uint32_t)job.input_len;
    const uint32_t cap = job.out_cap > kMaxPosB ? kMaxPosB : (uint32_t)job.out_cap;
    const uint64_t limit = job.output_limit;
    LZF_GLOBAL u32x4* const recs = (LZF_GLOBAL u32x4*)c.recs + sj.rec_off;
    for (uint32_t t = blockIdx.x; t < sj.ntile; t += gridDim.x) {
        __syncthreads();
        const uint32_t n = tile_enumerate(c, j, t, lane, len, in, stage, list);
        TileCtx tc{t * kSegTile, lds_addr(stage), len, in};
        const uint32_t tbase = c.tile_tok[(size_t)j * c.maxtile + t];
        uint32_t obase = c.tile_out[(size_t)j * c.maxtile + t];
        bool bad = false;
        for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
            const bool act = i0 + lane < n;
            Tok k; k.L = 0; k.M = 0; k.off = 0; k.src = 0; k.err = false;
            if (act) k = tile_decode(tc, tc.tstart + list[i0 + lane]);
            const uint32_t tot = k.L + k.M;
            const uint32_t incl = wave_scan_add(tot);
            const uint32_t lo = obase + (incl - tot), mo = lo + k.L;
            obase += __builtin_amdgcn_readlane(incl, 63);
            bool ok = act && !k.err;
            if (ok) {
                if (lo > cap || cap - lo < k.L) ok = false;                                   // our buffer (literals)
                else if (k.M && ((uint64_t)mo + k.M > limit || k.off == 0u || k.off > mo || cap - mo < k.M)) ok = false;   // :72-74, :83, :84-89, our buffer
            }
            bad = bad || (act && !ok);
            if (ok) {
                u32x4 r; r[0] = lo; r[1] = mo; r[2] = k.M; r[3] = k.off;
                recs[tbase + i0 + lane] = r;
                if (k.L > 0u && k.L <= 64u) copy_small_gg(out + lo, in + k.src, k.L);       // literals :65-67
                else if (k.L > 64u && k.L <= 256u) copy_medium_gg(out + lo, in + k.src, k.L);
            }
            for (unsigned long long m = __ballot(ok && k.L > 256u); m; m &= m - 1ull) {       // long runs: all lanes
                const uint32_t q = (uint32_t)__builtin_ctzll(m);
                wave_copy_long(out + __builtin_amdgcn_readlane(lo, q), in + __builtin_amdgcn_readlane(k.src, q), __builtin_amdgcn_readlane(k.L, q), lane);
            }
        }
        if (__ballot(bad) && lane == 0u) c.st[j].failed = 1u;
    }
}

// =====================================================================================================================
// levels: within a batch of 64 records, which matches copy from which
// =====================================================================================================================
// level(j) = 1 + max level of the matches (of the same batch) whose destination overlaps j's source bytes; 1 when there
// are none (everything in front of the batch is final when the batch starts).  Destinations are disjoint and in stream
// order, so the matches j depends on are a range of lanes [i_lo, i_hi], found by two binary searches; ranges wider than
// two lanes use the running maximum up to i_hi (never too small: a larger level is only later, not wrong).
__global__ __launch_bounds__(64) void lzf_seg_levels_kernel(seg_ctx c) {
    __shared__ uint32_t s_end[64], s_mo[64], s_lvl[64], s_pm[64];
    const uint32_t j = blockIdx.y;
    const seg_job sj = c.st[j];
    if (!sj.eligible || sj.failed) return;
    const uint32_t lane = threadIdx.x & 63u;
    LZF_GLOBAL u32x4* const recs = (LZF_GLOBAL u32x4*)c.recs + sj.rec_off;
    const uint32_t nb_all = (sj.ntok + 63u) / 64u;
    for (uint32_t b = blockIdx.x; b < nb_all; b += gridDim.x) {
        const uint32_t idx = b * 64u + lane;
        const bool act = idx < sj.ntok;
        u32x4 r = u32x4{0, 0, 0, 0};
        if (act) r = recs[idx];
        const uint32_t mo = r[1], M = r[2], off = r[3] & 0xFFFFu;
        const bool has = act && M != 0u;
        const uint32_t span = M < off ? M : off;
        const uint32_t s0 = mo - off, e0 = s0 + span;
        const uint32_t ob = __builtin_amdgcn_readlane(r[0], 0);
        __syncthreads();
        s_end[lane] = act ? mo + M : 0xFFFFFFFFu;
        s_mo[lane] = act ? mo : 0xFFFFFFFFu;
        uint32_t lvl = has ? 1u : 0u;
        const bool dep = has && e0 > ob;             // the source reaches into the batch
        uint32_t ilo = 0, ihi = 0; bool any_dep = false;
        if (__any(dep)) {
            __syncthreads();
            // ilo = #lanes with end <= s0; ihi = #lanes with mo < e0, minus 1
            uint32_t a = 0, bb = 0;
#pragma unroll
            for (uint32_t step = 32u; step; step >>= 1) {
                if (s_end[(a + step - 1u) & 63u] <= s0) a += step;
                if (s_mo[(bb + step - 1u) & 63u] < e0) bb += step;
            }
            // (a, bb in 0..63 by construction of the search; lane 63's end > s0 whenever dep)
            ilo = a; ihi = bb - 1u;
            any_dep = dep && bb >= 1u && ilo <= ihi && ihi < lane;
        }
        if (__any(any_dep)) {
            for (uint32_t it = 0; it < 64u; ++it) {
                __syncthreads();
                s_lvl[lane] = lvl;
                s_pm[lane] = wave_scan_max(lvl);
                __syncthreads();
                uint32_t nl = lvl;
                if (any_dep) {
                    const uint32_t m = ihi - ilo <= 1u ? (s_lvl[ilo] > s_lvl[ihi] ? s_lvl[ilo] : s_lvl[ihi]) : s_pm[ihi];
                    nl = 1u + m;
                }
                const bool ch = nl != lvl;
                lvl = nl;
                if (!__any(ch)) break;
            }
        }
        if (act) {
            const uint32_t flags = (has && M > 64u) ? kFlagCoop : 0u;
            recs[idx][3] = off | (lvl << 16) | (flags << 24);
        }
    }
}

// =====================================================================================================================
// resolve: the dependent match copies of a block, one LDS round per level
// =====================================================================================================================
// Biased positions y = x + rb (rb = out & 15): y % 16 == 0 <=> out + x is 16-byte aligned; ring index = y & (R - 1).
// The ring holds y in [max(fp - R, vlo), fp): fp = filled up to (whole granules, from `out`, where the literals already
// are); fl = flushed up to (everything below is final in HBM).
// Software pipeline, one batch of 64 records per iteration: the records of batch b + 2 and the granules of batch b + 1
// are loaded while batch b is resolved, so that no global round trip sits between two batches.
template <int R>
__global__ __launch_bounds__(64) void lzf_seg_resolve_kernel(seg_ctx c) {
    constexpr uint32_t kMask = (uint32_t)R - 1u;
    constexpr uint32_t kSpan = 4096;                   // output bytes one sub-batch may produce
    static_assert(R >= 32768, "far sources must be flushed: ring >= deferred flush (8 KiB) + two sub-batches + fetch-ahead (4 KiB)");
    __shared__ __attribute__((aligned(16))) uint8_t ring[R];
    const uint32_t j = blockIdx.x;
    if (j >= c.n_jobs) return;
    const seg_job sj = c.st[j];
    if (!sj.eligible || sj.failed) return;
    const uint32_t lane = threadIdx.x & 63u;
    const lzf_decompress_job job = c.jobs[j];
    const long long t_start = clock64();
    gu8* const out = as_global(job.out);
    const uint32_t rb = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u);
    gu8* const outb = out - rb;                        // outb + y = out + x
    const LZF_GLOBAL u32x4* const recs = (const LZF_GLOBAL u32x4*)c.recs + sj.rec_off;
    const uint32_t n = sj.ntok, total = sj.outb;
    const uint32_t ring_a = lds_addr(ring);
    uint32_t fp = 0, vlo = 0, fl = rb, safe = 0;       // safe: own stores below this y are visible to own loads
#ifdef LZF_SEG_TIME
    long long tm_rounds = 0, tm_pre = 0, tm_flush = 0; uint32_t n_rounds = 0;
#define SEGT(var) do { const long long t__ = clock64(); var += t__ - tm_t; tm_t = t__; } while (0)
    long long tm_t = clock64();
#else
#define SEGT(var) do { } while (0)
#endif

    // out[y0, y1) <- ring (biased positions)
    auto flush_range = [&](uint32_t y0, uint32_t y1) {
        if (y1 <= y0) return;
        uint32_t nh = (16u - (y0 & 15u)) & 15u; if (nh > y1 - y0) nh = y1 - y0;
        if (nh) { if (lane < nh) outb[y0 + lane] = ring[(y0 + lane) & kMask]; y0 += nh; }
        const uint32_t ng = (y1 - y0) >> 4;
        for (uint32_t g = lane; g < ng; g += kWave)
            *reinterpret_cast<LZF_GLOBAL u32x4*>(outb + y0 + 16u * g) = *reinterpret_cast<const u32x4*>(&ring[(y0 + 16u * g) & kMask]);
        y0 += ng << 4;
        if (lane < y1 - y0) outb[y0 + lane] = ring[(y0 + lane) & kMask];
    };
    // ring <- out, whole granules [fp, g1)
    auto fill_to = [&](uint32_t g1) {
        for (uint32_t y = fp + 16u * lane; y < g1; y += 16u * kWave)
            *reinterpret_cast<u32x4*>(&ring[y & kMask]) = *reinterpret_cast<const LZF_GLOBAL u32x4*>(outb + y);
        if (g1 > fp) fp = g1;
    };
    // ring[d, d + nbytes) <- ring[s, s + nbytes): source final, ranges disjoint; all lanes, 8 bytes each
    auto ring_copy = [&](uint32_t d, uint32_t s_, uint32_t nbytes) {
        const uint32_t n8 = nbytes >> 3;
        for (uint32_t u = lane; u < n8; u += kWave) {
            const uint32_t sa = (s_ + 8u * u) & kMask, da = (d + 8u * u) & kMask;
            if (sa + 8u <= (uint32_t)R && da + 8u <= (uint32_t)R) lds_st64(ring_a + da, lds_ld64u(ring_a + sa));
            else for (uint32_t t = 0; t < 8u; ++t) ring[(da + t) & kMask] = ring[(sa + t) & kMask];
        }
        const uint32_t t0 = n8 << 3;
        if (lane < nbytes - t0) ring[(d + t0 + lane) & kMask] = ring[(s_ + t0 + lane) & kMask];
    };
    // copy_overlapping (decompress.rs:80-138) inside the ring: out[d + t] = out[d - off + t], t < M.  The bytes already
    // copied double the source every step (a multiple of the period is a period).
    auto ring_match = [&](uint32_t d, uint32_t off_, uint32_t M_) {
        uint32_t pos = 0, av = off_;
        while (pos < M_) { const uint32_t cc = av < M_ - pos ? av : M_ - pos; ring_copy(d + pos, d - off_, cc); pos += cc; av += cc; }
    };

    // ---- prefetch registers: granules [pf0, pf1) of `out`, at most 4 KiB
    u32x4 pf[4] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
    uint32_t pf0 = 0, pf1 = 0;
    uint32_t flush_to = rb;                            // the ring is final below this; written to HBM one batch later
    u32x4 rA = u32x4{0, 0, 0, 0}, rB = u32x4{0, 0, 0, 0};
    if (lane < n) rA = recs[lane];
    if (64u + lane < n) rB = recs[64u + lane];

    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
        const uint32_t nb = n - i0 < 64u ? n - i0 : 64u;
        // Order of the memory operations of an iteration (one counter serves loads and stores, in issue order):
        //   use what was fetched one iteration ago -> stores of the previous batch -> fetches for the next iterations -> rounds.
        // ---- 1. the granules fetched for this batch go into the ring
        if (pf1 > pf0) {
            if (fp < pf0) fill_to(pf0);
            if (fp == pf0) {
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) { const uint32_t y = pf0 + 1024u * k + 16u * lane; if (y < pf1) *reinterpret_cast<u32x4*>(&ring[y & kMask]) = pf[k]; }
                fp = pf1;
            }
            pf0 = pf1 = 0;
        }
        // ---- 2. the previous batch goes to HBM
        if (flush_to > fl) { flush_range(fl, flush_to); fl = flush_to; }
        // ---- 3. this batch's records; fetch the granules of the next batch and the records of the one after
        const u32x4 r = rA;
        rA = rB;
        const uint32_t lo = r[0], mo = r[1], M = lane < nb ? r[2] : 0u, off = r[3] & 0xFFFFu, lvl = (r[3] >> 16) & 255u;
        const bool coopf = ((r[3] >> 24) & kFlagCoop) != 0u;
        const uint32_t endp = mo + M;
        {
            const uint32_t need0 = (__builtin_amdgcn_readlane(endp, (nb - 1u) & 63u) + rb + 15u) & ~15u;
            if (i0 + 64u < n) {
                const uint32_t nb1 = n - i0 - 64u < 64u ? n - i0 - 64u : 64u;
                const uint32_t need1 = (__builtin_amdgcn_readlane(rA[1] + rA[2], (nb1 - 1u) & 63u) + rb + 15u) & ~15u;
                const uint32_t from = fp > need0 ? fp : need0;
                uint32_t to = need1; if (to > from + 4096u) to = from + 4096u;
                if (to > from) {
                    pf0 = from; pf1 = to;
#pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) { const uint32_t y = from + 1024u * k + 16u * lane; if (y < to) pf[k] = *reinterpret_cast<const LZF_GLOBAL u32x4*>(outb + y); }
                }
            }
        }
        rB = u32x4{0, 0, 0, 0};
        if (i0 + 128u + lane < n) rB = recs[i0 + 128u + lane];
        SEGT(tm_pre);
        uint32_t a = 0;
        while (a < nb) {
            const uint32_t ob = __builtin_amdgcn_readlane(lo, a);
            const uint32_t bb0 = first_lane(__ballot(lane >= a && lane < nb && endp - ob > kSpan));
            const uint32_t b = bb0 < nb ? bb0 : nb;
            if (b == a) {
                // ---- a sequence larger than a sub-batch: its literals are in place; the match goes HBM -> HBM
                const uint32_t g_mo = __builtin_amdgcn_readlane(mo, a), g_M = __builtin_amdgcn_readlane(M, a), g_off = __builtin_amdgcn_readlane(off, a);
                const uint32_t g_lo = ob, g_end = g_mo + g_M;
                if (fp < ((g_lo + rb + 15u) & ~15u)) fill_to((g_lo + rb + 15u) & ~15u);   // (the ring up to the sequence, for the flush below)
                flush_range(fl, g_lo + rb);                                                 // everything in front of the sequence to HBM, byte-exact
                wave_store_fence();
                if (g_M) {
                    gu8* const src = out + (g_mo - g_off);
                    uint32_t pos = 0, av = g_off;
                    while (pos < g_M) {               // the same doubling as ring_match, through HBM
                        const uint32_t cc = av < g_M - pos ? av : g_M - pos;
                        wave_copy_long(out + g_mo + pos, src, cc, lane);
                        pos += cc; av += cc;
                        if (pos < g_M) wave_store_fence();
                    }
                    wave_store_fence();
                }
                // the ring starts again behind the sequence (the granule that holds its last bytes is read back by the next
                // fill; fl = ye keeps those bytes from being written twice — they are already there)
                const uint32_t ye = g_end + rb;
                vlo = ye & ~15u; fp = vlo; fl = ye; safe = ye; flush_to = ye;
                a += 1u;
                continue;
            }
            const bool inb = lane >= a && lane < b;
            const uint32_t oe = __builtin_amdgcn_readlane(endp, b - 1u);
            fill_to((oe + rb + 15u) & ~15u);
            const uint32_t lov = (fp > (uint32_t)R && fp - (uint32_t)R > vlo) ? fp - (uint32_t)R : vlo;   // the ring is valid from here
            const bool has = inb && M != 0u;
            const uint32_t span = M < off ? M : off;
            const uint32_t sy = mo - off + rb, dy = mo + rb;
            const bool near = has && sy >= lov;
            const bool far = has && sy + span <= lov;
            const bool mixed = has && !near && !far;
            const uint32_t di = dy & kMask, si = sy & kMask;
            const bool wrap = di + M > (uint32_t)R || si + M > (uint32_t)R;
            // ---- far sources: HBM -> ring (never overlapping: the distance exceeds a sub-batch)
            if (__ballot(far)) {
                if (__ballot(far && sy + span > safe)) { wave_store_fence(); safe = fl; }
                const bool far_own = far && M <= 64u && !wrap;
                if (far_own) put_small_glb(ring_a + di, outb + sy, M);
                for (unsigned long long m = __ballot(far && !far_own); m; m &= m - 1ull) {
                    const uint32_t q = (uint32_t)__builtin_ctzll(m);
                    const uint32_t qM = __builtin_amdgcn_readlane(M, q), qs = __builtin_amdgcn_readlane(sy, q), qd = __builtin_amdgcn_readlane(dy, q);
                    for (uint32_t i = lane; i < qM; i += kWave) ring[(qd + i) & kMask] = outb[qs + i];
                }
            }
            // ---- rounds.  Lanes that move their own match with one LDS round trip: A = 4..7 bytes (two 4-byte pieces),
            // B = 8..32 bytes (four 8-byte pieces, two-ended); both classes share the round trip.  The rest (33..64 bytes
            // or overlapping: own doubling steps; longer, wrapping, mixed sources: the whole wave) follows in the same round.
            const bool own = near && !coopf && !wrap && M <= 64u;
            const bool ovl = off < M;
            const bool fastA = own && !ovl && M < 8u, fastB = own && !ovl && M >= 8u && M <= 32u;
            const bool slow_own = own && !fastA && !fastB;
            const unsigned long long mA_all = __ballot(fastA), mB_all = __ballot(fastB);
            const unsigned long long mS_all = __ballot((near || mixed) && !fastA && !fastB);
            unsigned long long todo = mA_all | mB_all | mS_all;
            const uint32_t sa = ring_a + si, da = ring_a + di;
            const uint32_t o1 = M >= 16u ? 8u : M - 8u, o2 = M >= 16u ? M - 16u : 0u, o3 = M - 8u, a1 = M - 4u;
            for (uint32_t lv = 1; todo; ++lv) {
                const unsigned long long ml = __ballot(lvl == lv) & todo;
                if (!ml) { if (lv > 70u) break; continue; }
                todo &= ~ml;
#ifdef LZF_SEG_TIME
                ++n_rounds;
#endif
                const unsigned long long mA = ml & mA_all, mB = ml & mB_all, mS = ml & mS_all;
                if (mA | mB) {
                    uint32_t va0, va1; uint64_t vb0, vb1, vb2, vb3; unsigned long long sv;
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n\t"
                        "s_mov_b64 exec, %[mA]\n\t"
                        "ds_read_b32 %[va0], %[sa]\n\t"
                        "ds_read_b32 %[va1], %[sa1]\n\t"
                        "s_mov_b64 exec, %[mB]\n\t"
                        "ds_read_b64 %[vb0], %[sa]\n\t"
                        "ds_read_b64 %[vb1], %[sb1]\n\t"
                        "ds_read_b64 %[vb2], %[sb2]\n\t"
                        "ds_read_b64 %[vb3], %[sb3]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "ds_write_b64 %[da], %[vb0]\n\t"
                        "ds_write_b64 %[db1], %[vb1]\n\t"
                        "ds_write_b64 %[db2], %[vb2]\n\t"
                        "ds_write_b64 %[db3], %[vb3]\n\t"
                        "s_mov_b64 exec, %[mA]\n\t"
                        "ds_write_b32 %[da], %[va0]\n\t"
                        "ds_write_b32 %[da1], %[va1]\n\t"
                        "s_mov_b64 exec, %[sv]\n\t"
                        : [va0] "=&v"(va0), [va1] "=&v"(va1), [vb0] "=&v"(vb0), [vb1] "=&v"(vb1), [vb2] "=&v"(vb2), [vb3] "=&v"(vb3), [sv] "=&s"(sv)
                        : [mA] "s"(mA), [mB] "s"(mB), [sa] "v"(sa), [sa1] "v"(sa + a1), [sb1] "v"(sa + o1), [sb2] "v"(sa + o2), [sb3] "v"(sa + o3),
                          [da] "v"(da), [da1] "v"(da + a1), [db1] "v"(da + o1), [db2] "v"(da + o2), [db3] "v"(da + o3)
                        : "memory");
                }
                if (mS) {
                    const bool nowS = (mS >> lane) & 1ull;
                    if (nowS && slow_own) {                       // 33..64 bytes, or overlapping: doubling steps of its own
                        uint32_t pos = 0, av = off;
                        while (pos < M) { const uint32_t cc = av < M - pos ? av : M - pos; put_small_lds(da + pos, sa, cc); pos += cc; av += cc; }
                    }
                    for (unsigned long long m = __ballot(nowS && !slow_own); m; m &= m - 1ull) {
                        const uint32_t q = (uint32_t)__builtin_ctzll(m);
                        const uint32_t qM = __builtin_amdgcn_readlane(M, q), qs = __builtin_amdgcn_readlane(sy, q), qd = __builtin_amdgcn_readlane(dy, q);
                        const uint32_t qo = __builtin_amdgcn_readlane(off, q);
                        const bool qmixed = (__ballot(mixed) >> q) & 1ull;
                        if (qmixed) {
                            // source partly older than the ring: byte by byte, from HBM below lov
                            if (qs + (qo < qM ? qo : qM) > safe) { wave_store_fence(); safe = fl; }
                            if (lane == q) {
                                for (uint32_t i = 0; i < qM; ++i) {
                                    const uint32_t y = qs + (qo < qM ? i % qo : i);
                                    ring[(qd + i) & kMask] = y < lov ? (uint8_t)outb[y] : ring[y & kMask];
                                }
                            }
                        } else ring_match(qd, qo, qM);
                    }
                }
            }
            SEGT(tm_rounds);
            // ---- the sub-batch is final in the ring; it goes to HBM at the top of the next batch (or now, when much is pending)
            flush_to = (oe + rb) & ~15u;
            if (flush_to > fl && flush_to - fl > 8192u) { flush_range(fl, flush_to); fl = flush_to; }
            SEGT(tm_flush);
            a = b;
        }
    }
    if (flush_to > fl) { flush_range(fl, flush_to); fl = flush_to; }
    flush_range(fl, total + rb);
    if (lane == 0u) {
        c.results[j].out_len = total;
        c.results[j].status = LZF_OK;
        c.results[j].reserved = (uint32_t)((clock64() - t_start) >> 10);
#ifdef LZF_SEG_TIME
        c.st[j].pad = n_rounds;
        c.st[j].pad2 = (uint64_t)(uint32_t)(tm_rounds >> 10) | ((uint64_t)(uint16_t)(tm_pre >> 14) << 32) | ((uint64_t)(uint16_t)(tm_flush >> 14) << 48);
#endif
        c.st[j].done = 1u;
    }
#undef SEGT
}

// =====================================================================================================================
// resolve, as a pair of wavefronts per block: the STAGER (wave 1) does everything that is not the dependency chain — the
// ring's granules in from `out` (literals are there), finished granules out to HBM, the few sources that are older than the
// ring (read back from HBM), sequences larger than a sub-batch (HBM -> HBM), and per lane the LDS addresses, length class
// and level of its match, parked in an LDS slot; the RESOLVER (wave 0) takes slot after slot and runs the rounds.
// Tickets: the stager publishes sub-batch t with ctl[0] = t + 1 after its LDS writes; the resolver answers with
// ctl[1] = t + 1 after its own.  LDS executes one wave's accesses in order, so a flag is never seen before the data.
// =====================================================================================================================
template <int R>
__global__ __launch_bounds__(128) void lzf_seg_resolve_pair_kernel(seg_ctx c) {
    constexpr uint32_t kMask = (uint32_t)R - 1u;
    constexpr uint32_t kSpan = (uint32_t)R / 8u;       // output bytes one sub-batch may produce
    constexpr uint32_t NS = 3;                         // slots: sub-batches the stager may be ahead
    constexpr uint32_t kAhead = NS * kSpan + 64u;      // ... i.e. the fill pointer may be this far beyond a sub-batch when it is resolved
    __shared__ __attribute__((aligned(16))) uint8_t ring[R];
    __shared__ __attribute__((aligned(16))) u32x4 slots[NS][64];
    __shared__ uint32_t ctl[4];                        // [0] staged tickets, [1] resolved tickets, [2] stager finished
    const uint32_t j = blockIdx.x;
    if (j >= c.n_jobs) return;
    const seg_job sj = c.st[j];
    if (!sj.eligible || sj.failed) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const lzf_decompress_job job = c.jobs[j];
    const long long t_start = clock64();
    gu8* const out = as_global(job.out);
    const uint32_t rb = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u);
    gu8* const outb = out - rb;
    const LZF_GLOBAL u32x4* const recs = (const LZF_GLOBAL u32x4*)c.recs + sj.rec_off;
    const uint32_t n = sj.ntok, total = sj.outb;
    const uint32_t ring_a = lds_addr(ring);
    volatile uint32_t* const vctl = ctl;
    if (threadIdx.x < 4u) ctl[threadIdx.x] = 0u;
    __syncthreads();

    // ring[d, d + nbytes) <- ring[s, s + nbytes): source final, ranges disjoint; all lanes, 8 bytes each
    auto ring_copy = [&](uint32_t d, uint32_t s_, uint32_t nbytes) {
        const uint32_t n8 = nbytes >> 3;
        for (uint32_t u = lane; u < n8; u += kWave) {
            const uint32_t sa = (s_ + 8u * u) & kMask, da = (d + 8u * u) & kMask;
            if (sa + 8u <= (uint32_t)R && da + 8u <= (uint32_t)R) lds_st64(ring_a + da, lds_ld64u(ring_a + sa));
            else for (uint32_t t = 0; t < 8u; ++t) ring[(da + t) & kMask] = ring[(sa + t) & kMask];
        }
        const uint32_t t0 = n8 << 3;
        if (lane < nbytes - t0) ring[(d + t0 + lane) & kMask] = ring[(s_ + t0 + lane) & kMask];
    };
    auto ring_match = [&](uint32_t d, uint32_t off_, uint32_t M_) {
        uint32_t pos = 0, av = off_;
        while (pos < M_) { const uint32_t cc = av < M_ - pos ? av : M_ - pos; ring_copy(d + pos, d - off_, cc); pos += cc; av += cc; }
    };

    if (role == 0u) {
        // ================================================ RESOLVER ================================================
        for (uint32_t t = 0;; ++t) {
            uint32_t staged;
            for (;;) {
                staged = vctl[0];
                if (staged > t) break;
                if (vctl[2]) { staged = vctl[0]; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (staged <= t) break;
            const u32x4 d = slots[t % NS][lane];
            // d[0] = LDS address of the source, d[1] = of the destination, d[2] = M | lvl << 16 | class << 24, d[3] = off | lanes << 16 (lane 0..)
            const uint32_t sa = d[0], da = d[1], M = d[2] & 0xFFFFu, lvl = (d[2] >> 16) & 255u, cls = d[2] >> 24, off = d[3];
            // class: 1 = A (4..7), 2 = B (8..32), 3 = C (33..64), 4 = own doubling steps (overlapping, <= 64), 5 = whole wave; 0 = nothing to do
            const unsigned long long mA_all = __ballot(cls == 1u), mB_all = __ballot(cls == 2u), mC_all = __ballot(cls == 3u);
            const unsigned long long mS_all = __ballot(cls >= 4u);
            unsigned long long todo = mA_all | mB_all | mC_all | mS_all;
            const uint32_t o1 = M >= 16u ? 8u : M - 8u, o2 = M >= 16u ? M - 16u : 0u, o3 = M - 8u, a1 = M - 4u;
            for (uint32_t lv = 1; todo; ++lv) {
                const unsigned long long ml = __ballot(lvl == lv) & todo;
                if (!ml) { if (lv > 70u) break; continue; }
                todo &= ~ml;
                const unsigned long long mA = ml & mA_all, mB = ml & (mB_all | mC_all), mC = ml & mC_all, mS = ml & mS_all;
                if (mA | mB) {
                    uint32_t va0, va1; uint64_t vb0, vb1, vb2, vb3, vc0, vc1, vc2, vc3; unsigned long long sv;
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n\t"
                        "s_mov_b64 exec, %[mA]\n\t"
                        "ds_read_b32 %[va0], %[sa]\n\t"
                        "ds_read_b32 %[va1], %[sa1]\n\t"
                        "s_mov_b64 exec, %[mB]\n\t"
                        "ds_read_b64 %[vb0], %[sa]\n\t"
                        "ds_read_b64 %[vb1], %[sb1]\n\t"
                        "ds_read_b64 %[vb2], %[sb2]\n\t"
                        "ds_read_b64 %[vb3], %[sb3]\n\t"
                        "s_mov_b64 exec, %[mC]\n\t"
                        "s_cbranch_execz Lnc1%=\n\t"
                        "ds_read_b64 %[vc0], %[sa] offset:16\n\t"
                        "ds_read_b64 %[vc1], %[sa] offset:24\n\t"
                        "ds_read_b64 %[vc2], %[sc2]\n\t"
                        "ds_read_b64 %[vc3], %[sc3]\n\t"
                        "Lnc1%=:\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "s_cbranch_execz Lnc2%=\n\t"
                        "ds_write_b64 %[da], %[vc0] offset:16\n\t"
                        "ds_write_b64 %[da], %[vc1] offset:24\n\t"
                        "ds_write_b64 %[dc2], %[vc2]\n\t"
                        "ds_write_b64 %[dc3], %[vc3]\n\t"
                        "Lnc2%=:\n\t"
                        "s_mov_b64 exec, %[mB]\n\t"
                        "ds_write_b64 %[da], %[vb0]\n\t"
                        "ds_write_b64 %[db1], %[vb1]\n\t"
                        "ds_write_b64 %[db2], %[vb2]\n\t"
                        "ds_write_b64 %[db3], %[vb3]\n\t"
                        "s_mov_b64 exec, %[mA]\n\t"
                        "ds_write_b32 %[da], %[va0]\n\t"
                        "ds_write_b32 %[da1], %[va1]\n\t"
                        "s_mov_b64 exec, %[sv]\n\t"
                        : [va0] "=&v"(va0), [va1] "=&v"(va1), [vb0] "=&v"(vb0), [vb1] "=&v"(vb1), [vb2] "=&v"(vb2), [vb3] "=&v"(vb3),
                          [vc0] "=&v"(vc0), [vc1] "=&v"(vc1), [vc2] "=&v"(vc2), [vc3] "=&v"(vc3), [sv] "=&s"(sv)
                        : [mA] "s"(mA), [mB] "s"(mB), [mC] "s"(mC), [sa] "v"(sa), [sa1] "v"(sa + a1), [sb1] "v"(sa + o1), [sb2] "v"(sa + o2), [sb3] "v"(sa + o3),
                          [sc2] "v"(sa + M - 32u), [sc3] "v"(sa + M - 24u),
                          [da] "v"(da), [da1] "v"(da + a1), [db1] "v"(da + o1), [db2] "v"(da + o2), [db3] "v"(da + o3), [dc2] "v"(da + M - 32u), [dc3] "v"(da + M - 24u)
                        : "memory");
                }
                if (mS) {
                    const bool nowS = (mS >> lane) & 1ull;
                    if (nowS && cls == 4u) {                      // overlapping, at most 64 bytes: doubling steps of its own
                        uint32_t pos = 0, av = off;
                        while (pos < M) { const uint32_t cc = av < M - pos ? av : M - pos; put_small_lds(da + pos, sa, cc); pos += cc; av += cc; }
                    }
                    for (unsigned long long m = __ballot(nowS && cls == 5u); m; m &= m - 1ull) {
                        const uint32_t q = (uint32_t)__builtin_ctzll(m);
                        // (whole-wave matches carry their biased position in sa / da and the full length in off's slot)
                        ring_match(__builtin_amdgcn_readlane(da, q), __builtin_amdgcn_readlane(off & 0xFFFFu, q), __builtin_amdgcn_readlane(sa, q));
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0u) vctl[1] = t + 1u;
        }
    } else {
        // ================================================= STAGER =================================================
        uint32_t fp = 0, vlo = 0, fl = rb, safe = 0;
        uint32_t ticket = 0;                             // sub-batches published
        uint32_t endq[NS];                               // biased end (rounded down to a granule) of the published sub-batches, by slot
#pragma unroll
        for (uint32_t i = 0; i < NS; ++i) endq[i] = rb;
        uint32_t flushed_t = 0;                          // tickets whose bytes are in HBM
        auto flush_range = [&](uint32_t y0, uint32_t y1) {
            if (y1 <= y0) return;
            uint32_t nh = (16u - (y0 & 15u)) & 15u; if (nh > y1 - y0) nh = y1 - y0;
            if (nh) { if (lane < nh) outb[y0 + lane] = ring[(y0 + lane) & kMask]; y0 += nh; }
            const uint32_t ng = (y1 - y0) >> 4;
            for (uint32_t g = lane; g < ng; g += kWave)
                *reinterpret_cast<LZF_GLOBAL u32x4*>(outb + y0 + 16u * g) = *reinterpret_cast<const u32x4*>(&ring[(y0 + 16u * g) & kMask]);
            y0 += ng << 4;
            if (lane < y1 - y0) outb[y0 + lane] = ring[(y0 + lane) & kMask];
        };
        auto fill_to = [&](uint32_t g1) {
            for (uint32_t y = fp + 16u * lane; y < g1; y += 16u * kWave)
                *reinterpret_cast<u32x4*>(&ring[y & kMask]) = *reinterpret_cast<const LZF_GLOBAL u32x4*>(outb + y);
            if (g1 > fp) fp = g1;
        };
        // write what the resolver has finished to HBM (whole granules), in ticket order
        auto flush_resolved = [&](uint32_t upto_t) {
            while (flushed_t < upto_t) {
                uint32_t e = endq[0];
#pragma unroll
                for (uint32_t i = 1; i < NS; ++i) if (flushed_t % NS == i) e = endq[i];
                if (e > fl) { flush_range(fl, e); fl = e; }
                ++flushed_t;
            }
        };
        auto wait_resolved = [&](uint32_t t) { while (vctl[1] < t) __builtin_amdgcn_s_sleep(1); };

        u32x4 rA = u32x4{0, 0, 0, 0}, rB = u32x4{0, 0, 0, 0};
        if (lane < n) rA = recs[lane];
        if (64u + lane < n) rB = recs[64u + lane];
        for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
            const uint32_t nb = n - i0 < 64u ? n - i0 : 64u;
            const u32x4 r = rA;
            rA = rB;
            rB = u32x4{0, 0, 0, 0};
            if (i0 + 128u + lane < n) rB = recs[i0 + 128u + lane];
            const uint32_t lo = r[0], mo = r[1], M = lane < nb ? r[2] : 0u, off = r[3] & 0xFFFFu, lvl = (r[3] >> 16) & 255u;
            const uint32_t endp = mo + M;
            uint32_t a = 0;
            while (a < nb) {
                const uint32_t ob = __builtin_amdgcn_readlane(lo, a);
                const uint32_t bb0 = first_lane(__ballot(lane >= a && lane < nb && endp - ob > kSpan));
                const uint32_t b = bb0 < nb ? bb0 : nb;
                if (b == a) {
                    // ---- a sequence larger than a sub-batch: everything in front of it resolved and in HBM, then HBM -> HBM
                    const uint32_t g_mo = __builtin_amdgcn_readlane(mo, a), g_M = __builtin_amdgcn_readlane(M, a), g_off = __builtin_amdgcn_readlane(off, a);
                    const uint32_t g_lo = ob, g_end = g_mo + g_M;
                    wait_resolved(ticket);
                    flush_resolved(ticket);
                    if (fp < ((g_lo + rb + 15u) & ~15u)) fill_to((g_lo + rb + 15u) & ~15u);
                    flush_range(fl, g_lo + rb);
                    wave_store_fence();
                    if (g_M) {
                        gu8* const src = out + (g_mo - g_off);
                        uint32_t pos = 0, av = g_off;
                        while (pos < g_M) {
                            const uint32_t cc = av < g_M - pos ? av : g_M - pos;
                            wave_copy_long(out + g_mo + pos, src, cc, lane);
                            pos += cc; av += cc;
                            if (pos < g_M) wave_store_fence();
                        }
                        wave_store_fence();
                    }
                    const uint32_t ye = g_end + rb;
                    vlo = ye & ~15u; fp = vlo; fl = ye; safe = ye;
                    a += 1u;
                    continue;
                }
                // ---- sub-batch [a, b): a free slot, the ring filled, classes, sources older than the ring
                if (ticket >= NS) { wait_resolved(ticket - NS + 1u); }
                flush_resolved(vctl[1] < ticket ? vctl[1] : ticket);
                const bool inb = lane >= a && lane < b;
                const uint32_t oe = __builtin_amdgcn_readlane(endp, b - 1u);
                fill_to((oe + rb + 15u) & ~15u);
                // when the resolver gets here the fill pointer may be kAhead further: what the ring still holds then
                const uint32_t fpw = fp + kAhead;
                const uint32_t lov = (fpw > (uint32_t)R && fpw - (uint32_t)R > vlo) ? fpw - (uint32_t)R : vlo;
                const bool has = inb && M != 0u;
                const uint32_t span = M < off ? M : off;
                const uint32_t sy = mo - off + rb, dy = mo + rb;
                const bool near = has && sy >= lov;
                const uint32_t di = dy & kMask, si = sy & kMask;
                const bool wrap = di + M > (uint32_t)R || si + M > (uint32_t)R;
                // sources (partly) older than that: moved here, from HBM below the flushed mark and from the ring above it
                // (those ring bytes are older than every sub-batch in flight: final)
                if (__ballot(has && !near)) {
                    if (__ballot(has && !near && sy + span > fl)) { wait_resolved(ticket); flush_resolved(ticket); }
                    if (__ballot(has && !near && (sy + span > fl ? fl : sy + span) > safe)) { wave_store_fence(); safe = fl; }
                    const bool far_own = has && !near && sy + span <= fl && M <= 64u && !wrap && off >= M;
                    if (far_own) put_small_glb(ring_a + di, outb + sy, M);
                    for (unsigned long long m = __ballot(has && !near && !far_own); m; m &= m - 1ull) {
                        const uint32_t q = (uint32_t)__builtin_ctzll(m);
                        const uint32_t qM = __builtin_amdgcn_readlane(M, q), qs = __builtin_amdgcn_readlane(sy, q), qd = __builtin_amdgcn_readlane(dy, q);
                        const uint32_t qo = __builtin_amdgcn_readlane(off, q);
                        if (lane == q) {
                            for (uint32_t i = 0; i < qM; ++i) {
                                const uint32_t y = qs + (qo < qM ? i % qo : i);
                                ring[(qd + i) & kMask] = y < fl ? (uint8_t)outb[y] : ring[y & kMask];
                            }
                        }
                    }
                }
                uint32_t cls = 0;
                if (near) {
                    if (M > 64u || wrap) cls = 5u;
                    else if (off < M) cls = 4u;
                    else cls = M < 8u ? 1u : M <= 32u ? 2u : 3u;
                }
                u32x4 d;
                d[0] = cls == 5u ? M : ring_a + si;          // whole-wave matches: length / biased destination / offset
                d[1] = cls == 5u ? dy : ring_a + di;
                d[2] = (M & 0xFFFFu) | (lvl << 16) | (cls << 24);
                d[3] = off;
                slots[ticket % NS][lane] = d;
                const uint32_t eg = (oe + rb) & ~15u;
#pragma unroll
                for (uint32_t i = 0; i < NS; ++i) if (ticket % NS == i) endq[i] = eg;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                ++ticket;
                if (lane == 0u) vctl[0] = ticket;
                a = b;
            }
        }
        wait_resolved(ticket);
        flush_resolved(ticket);
        flush_range(fl, total + rb);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lane == 0u) {
            vctl[2] = 1u;
            c.results[j].out_len = total;
            c.results[j].status = LZF_OK;
            c.results[j].reserved = (uint32_t)((clock64() - t_start) >> 10);
            c.st[j].done = 1u;
        }
    }
}
template __global__ void lzf_seg_resolve_pair_kernel<32768>(seg_ctx);
template __global__ void lzf_seg_resolve_pair_kernel<65536>(seg_ctx);
template __global__ void lzf_seg_resolve_pair_kernel<131072>(seg_ctx);

template __global__ void lzf_seg_resolve_kernel<32768>(seg_ctx);
template __global__ void lzf_seg_resolve_kernel<65536>(seg_ctx);
template __global__ void lzf_seg_resolve_kernel<131072>(seg_ctx);

}  // namespace lzf
