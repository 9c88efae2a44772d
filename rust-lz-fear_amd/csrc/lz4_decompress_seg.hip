// lz4_decompress_seg.hip — raw::decompress_raw (src/raw/decompress.rs:58-138) for gfx950 with ONE BLOCK DECODED BY MANY
// WAVEFRONTS: the segmented pipeline of kernels.h (plan, parse, seam, tilesum, scan, records + levels, resolve).
//
// Why: a block is a serial chain twice over — the position of a token depends on every token before it
// (decompress.rs:61-71), and a match copies bytes an earlier match produced (:80-138; on text the chain of dependent
// matches is ~1/11 of the sequences long in windows of 64, tools/seg_depth.c).  One workgroup per block therefore needs ~20 ms per 4 MiB
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
constexpr uint32_t kCB = kSegChunk;                // staged bytes of a chunk (with the rows: 20 480 bytes of LDS, eight chunks per CU; tokens that reach over the end take the general routine)
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
    s.eligible = ((c.fed || (job.prefix_len == 0 && job.out_existing_len == 0)) && job.input_len >= c.min_in && job.input_len <= c.max_in &&
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
    // one array: rows A, rows B, the chunk's bytes.  A hop reads 4 bytes from position - 1: for the chunk's first byte that is the last
    // byte of row B (any value: a walk's first hop does not look at it), and no hop lands on the last two bytes (see `fe`).
    __shared__ __attribute__((aligned(16))) uint8_t lds_parse[2u * 64u * 8u * 4u + kCB];
    static_assert(sizeof(lds_parse) == 20480, "eight chunks per CU: an eighth of 160 KiB each");
    uint32_t* const rowsA = reinterpret_cast<uint32_t*>(lds_parse);
    uint32_t* const rowsB = rowsA + 64u * 8u;
    const uint32_t j = seg_job_of(c, blockIdx.y);
    const seg_job sj = c.st[j];
    if (!sj.eligible) return;
    const uint32_t lane = threadIdx.x & 63u;
    const lzf_decompress_job job = c.jobs[j];
    cgu8* __restrict__ in = as_global(job.input);
    const uint32_t len = (uint32_t)job.input_len;
    uint8_t* const cbuf = lds_parse + 2u * 64u * 8u * 4u;
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
        }
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) { ROWA(i) = 0u; ROWB(i) = 0u; }
        const uint32_t room = len - cstart;                                      // bytes from the chunk start to the end of the input
        // a plain hop lands below fe (chunk-relative): 24 bytes short of the input's end, and 2 short of the staged bytes — a lane reads the
        // 4 bytes from its position - 1 wherever it has landed (the extension byte of the token before is checked there), and a read
        // that crosses the end of the workgroup's LDS returns zeros for all four
        const uint32_t fe = room > 24u ? (room - 24u < kCB - 2u ? room - 24u : kCB - 2u) : 0u;
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
#ifndef LZF_SEG_NOASM
                // ---- plain hops, hand-scheduled (the walks are chains of dependent instructions: every one counts; hipcc's version of
                //      this loop spends ~20 scalar instructions per hop on lane masks).  A lane is in the loop while it is in EXEC and leaves it
                //      for good when its hop is not a plain one: past the stop position, parked for the general routine, on a token marked in
                //      row A (mode 1), or because the previous token's match length turned out to go on (0xFF extension byte: r steps back
                //      to that token, which the general routine then takes).  21 vector instructions per hop (mode 1: 26; 34 / 40 with the
                //      lanes' state in selects instead of EXEC).  mxp = the byte in front of the token that means "not plain": 0xFF when the
                //      previous token's match nibble was 15, else 0x100 (no byte has that value).
                {
                    uint32_t mxp = 0x100u, rprev = r, mg = merged ? 1u : 0u;
                    const uint32_t stopv = stop;
                    const unsigned long long m0 = __ballot(go && !merged && r < stop);
                    uint32_t a_, w_, t_, x_, rn_, mk_;
                    unsigned long long sv, sc;
                    // row word of position r: rows[(r >> 5) - 8 * lane][lane] = rowsA + ((r >> 5) << 8) + 4 * lane - (lane << 11)  (mod 2^32)
                    const uint32_t rowa2 = lds_addr(rowsA) + lane * 4u - (lane << 11), rowd = lds_addr(rowsB) - lds_addr(rowsA);
                    const uint32_t vzero = 0u, vff = 0xFFu, v100 = 0x100u;
                    if (mode == 0) {
                        asm volatile(
                            "s_mov_b64 %[sv], exec\n\t"
                            "s_and_b64 exec, exec, %[m0]\n\t"
                            "s_cbranch_execz Ld%=\n"
                            "Lw%=:\n\t"
                            "v_add_u32 %[a], %[cbm1], %[r]\n\t"
                            "ds_read_b32 %[w], %[a]\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "v_cmp_eq_u32_sdwa vcc, %[w], %[mxp] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
                            "v_cndmask_b32 %[r], %[r], %[rprev], vcc\n\t"
                            "s_andn2_b64 exec, exec, vcc\n\t"
                            "v_cmpx_lt_u32 vcc, %[r], %[stop]\n\t"
                            "v_bfe_u32 %[x], %[w], 12, 4\n\t"
                            "v_cmp_eq_u32 vcc, 15, %[x]\n\t"
                            "v_cndmask_b32_sdwa %[t], %[vz], %[w], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
                            "v_addc_co_u32 %[rn], vcc, %[x], %[t], vcc\n\t"
                            "v_add3_u32 %[rn], %[rn], %[r], 3\n\t"
                            "v_and_b32 %[t], 0xfff000, %[w]\n\t"
                            "v_cmpx_ne_u32 vcc, 0xfff000, %[t]\n\t"
                            "v_and_b32 %[x], 0xf00, %[w]\n\t"
                            "v_cmp_eq_u32 vcc, 0xf00, %[x]\n\t"
                            "v_addc_co_u32 %[rn], %[sc], 0, %[rn], vcc\n\t"
                            "v_cndmask_b32 %[mxp], %[v100], %[vff], vcc\n\t"
                            "v_cmpx_gt_u32 vcc, %[fe], %[rn]\n\t"
                            "v_lshrrev_b32 %[x], 5, %[r]\n\t"
                            "v_lshl_add_u32 %[x], %[x], 8, %[rowa2]\n\t"
                            "v_lshlrev_b32 %[a], %[r], 1\n\t"
                            "ds_or_b32 %[x], %[a]\n\t"
                            "v_mov_b32 %[rprev], %[r]\n\t"
                            "v_mov_b32 %[r], %[rn]\n\t"
                            "s_cbranch_execnz Lw%=\n"
                            "Ld%=:\n\t"
                            "s_mov_b64 exec, %[sv]"
                            : [r] "+v"(r), [rprev] "+v"(rprev), [mxp] "+v"(mxp),
                              [a] "=&v"(a_), [w] "=&v"(w_), [t] "=&v"(t_), [x] "=&v"(x_), [rn] "=&v"(rn_), [sv] "=&s"(sv), [sc] "=&s"(sc)
                            : [cbm1] "s"(cbuf_a - 1u), [fe] "s"(fe), [stop] "v"(stopv), [rowa2] "v"(rowa2), [vz] "v"(vzero), [vff] "v"(vff), [v100] "v"(v100), [m0] "s"(m0)
                            : "vcc", "memory");
                    } else {
                        asm volatile(
                            "s_mov_b64 %[sv], exec\n\t"
                            "s_and_b64 exec, exec, %[m0]\n\t"
                            "s_cbranch_execz Ld%=\n"
                            "Lw%=:\n\t"
                            "v_add_u32 %[a], %[cbm1], %[r]\n\t"
                            "v_lshrrev_b32 %[x], 5, %[r]\n\t"
                            "ds_read_b32 %[w], %[a]\n\t"
                            "v_lshl_add_u32 %[x], %[x], 8, %[rowa2]\n\t"
                            "ds_read_b32 %[mk], %[x]\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "v_cmp_eq_u32_sdwa vcc, %[w], %[mxp] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
                            "v_cndmask_b32 %[r], %[r], %[rprev], vcc\n\t"
                            "s_andn2_b64 exec, exec, vcc\n\t"
                            "v_cmpx_lt_u32 vcc, %[r], %[stop]\n\t"
                            "v_lshrrev_b32 %[mk], %[r], %[mk]\n\t"
                            "v_and_b32 %[mk], 1, %[mk]\n\t"
                            "v_or_b32 %[mg], %[mg], %[mk]\n\t"                 /* on a token of the pass-0 chain: merged */
                            "v_cmpx_eq_u32 vcc, 0, %[mk]\n\t"
                            "v_bfe_u32 %[mk], %[w], 12, 4\n\t"
                            "v_cmp_eq_u32 vcc, 15, %[mk]\n\t"
                            "v_cndmask_b32_sdwa %[t], %[vz], %[w], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
                            "v_addc_co_u32 %[rn], vcc, %[mk], %[t], vcc\n\t"
                            "v_add3_u32 %[rn], %[rn], %[r], 3\n\t"
                            "v_and_b32 %[t], 0xfff000, %[w]\n\t"
                            "v_cmpx_ne_u32 vcc, 0xfff000, %[t]\n\t"
                            "v_and_b32 %[mk], 0xf00, %[w]\n\t"
                            "v_cmp_eq_u32 vcc, 0xf00, %[mk]\n\t"
                            "v_addc_co_u32 %[rn], %[sc], 0, %[rn], vcc\n\t"
                            "v_cndmask_b32 %[mxp], %[v100], %[vff], vcc\n\t"
                            "v_cmpx_gt_u32 vcc, %[fe], %[rn]\n\t"
                            "v_add_u32 %[x], %[x], %[rowd]\n\t"                 /* the same word of row B */
                            "v_lshlrev_b32 %[a], %[r], 1\n\t"
                            "ds_or_b32 %[x], %[a]\n\t"
                            "v_mov_b32 %[rprev], %[r]\n\t"
                            "v_mov_b32 %[r], %[rn]\n\t"
                            "s_cbranch_execnz Lw%=\n"
                            "Ld%=:\n\t"
                            "s_mov_b64 exec, %[sv]"
                            : [r] "+v"(r), [rprev] "+v"(rprev), [mxp] "+v"(mxp), [mg] "+v"(mg),
                              [a] "=&v"(a_), [w] "=&v"(w_), [t] "=&v"(t_), [x] "=&v"(x_), [rn] "=&v"(rn_), [mk] "=&v"(mk_), [sv] "=&s"(sv), [sc] "=&s"(sc)
                            : [cbm1] "s"(cbuf_a - 1u), [fe] "s"(fe), [stop] "v"(stopv), [rowa2] "v"(rowa2), [rowd] "v"(rowd), [vz] "v"(vzero), [vff] "v"(vff), [v100] "v"(v100), [m0] "s"(m0)
                            : "vcc", "memory");
                    }
                    merged = mg != 0u;
                }
#else
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
#endif
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
    if (blockIdx.x >= c.g_n) return;
    const uint32_t j = c.grouped ? seg_job_of(c, blockIdx.x) : blockIdx.x;
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
                            // the common token — lengths with at most one extension byte, everything staged, not the input's end — from
                            // one unaligned LDS word (+ one byte): the position token_next_gen returns, at a quarter of its instructions
                            {
                                uint32_t w;
                                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(wb_a + (p - ws)) : "memory");
                                const uint32_t L0 = (w >> 4) & 15u, M0 = w & 15u, b1 = (w >> 8) & 255u;
                                const uint32_t q2 = p + 1u + (L0 == 15u ? 1u + 15u + b1 : L0);       // the offset's position
                                if (!(L0 == 15u && b1 == 255u) && q2 + 3u <= len && q2 - ws + 3u <= kWin + 128u &&
                                    !(M0 == 15u && wbytes[q2 + 2u - ws] == 255u)) { p = q2 + 2u + (M0 == 15u ? 1u : 0u); continue; }
                            }
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
// WANT_OFF = false: the lengths alone (the tile sums do not look at the offset).
template <bool WANT_OFF = true>
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
    if (WANT_OFF) k.off = tile_rdb(t, q) | (tile_rdb(t, q + 1u) << 8);
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
    const uint32_t j = seg_job_of(c, blockIdx.y);
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
                const Tok k = tile_decode<false>(tc, tc.tstart + list[i0 + lane]);
                err = err || k.err;
                sum += k.L + k.M;                          // (each <= 2^26 + 4, at most 11 per lane)
            }
        }
        const uint64_t tot = (uint64_t)__builtin_amdgcn_readlane(wave_scan_add(sum & 0xFFFFu), 63) +
                             ((uint64_t)__builtin_amdgcn_readlane(wave_scan_add(sum >> 16), 63) << 16);
        if (tot > 0x7FFFFFFFull) err = true;
        // (records are kept in batches of 64 that never span tiles: a tile's tokens take ceil(n / 64) batches, the last one padded)
        if (lane == 0u) { c.tile_tok[(size_t)j * c.maxtile + t] = (n + 63u) / 64u; c.tile_out[(size_t)j * c.maxtile + t] = (uint32_t)tot; }
        if (__ballot(err) && lane == 0u) c.st[j].failed = 1u;
    }
}

__global__ __launch_bounds__(64) void lzf_seg_scan_kernel(seg_ctx c) {
    if (blockIdx.x >= c.g_n) return;
    const uint32_t j = c.grouped ? seg_job_of(c, blockIdx.x) : blockIdx.x;
    const seg_job sj = c.st[j];
    if (!sj.eligible || sj.failed) return;
    const uint32_t lane = threadIdx.x & 63u;
    const lzf_decompress_job job = c.jobs[j];
    LZF_GLOBAL uint32_t* const TT = (LZF_GLOBAL uint32_t*)c.tile_tok + (size_t)j * c.maxtile;
    LZF_GLOBAL uint32_t* const TO = (LZF_GLOBAL uint32_t*)c.tile_out + (size_t)j * c.maxtile;
    uint64_t ctok = 0, cout = 0;
    // (eight steps' loads in flight at once: with one step per round trip to HBM the kernel was 33 dependent loads long, 75 us)
    for (uint32_t t8 = 0; t8 < sj.ntile; t8 += 512u) {
        uint32_t av[8], bv[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) { const uint32_t t = t8 + 64u * k + lane; av[k] = t < sj.ntile ? TT[t] : 0u; bv[k] = t < sj.ntile ? TO[t] : 0u; }
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            const uint32_t t = t8 + 64u * k + lane;
            const uint32_t a = av[k], b = bv[k];
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
    }
    if (lane == 0u) {
        const uint64_t cap = job.out_cap > kMaxPosB ? kMaxPosB : job.out_cap;
        ctok *= 64ull;                                                           // batches -> records (padded)
        bool ok = cout <= cap && ctok < 0x7FFFFFFFull;
        uint64_t off = 0;
        if (ok) {
            const uint64_t need = ctok + 64ull;
            off = atomicAdd(c.rec_top, (unsigned long long)need);
            if (off + need > c.rec_cap) ok = false;
        }
        if (!ok) c.st[j].failed = 1u;
        else { c.st[j].ntok = (uint32_t)ctok; c.st[j].outb = (uint32_t)cout; c.st[j].rec_off = off; }
    }
}

// =====================================================================================================================
// records + literals + levels: everything the resolve stage needs to know about a sequence, so that its two wavefronts
// compute nothing of it themselves
// =====================================================================================================================
// Per batch of 64 tokens of a tile (batches never span tiles; the last one of a tile is padded with empty sequences):
// decode (decompress.rs:61-74), absolute positions, the checks of :72-74,:83-89, LITERALS -> out, then
// level(j) = 1 + max level of the matches (of the same batch) whose destination overlaps j's source bytes; 1 when there
// are none (everything in front of the batch is final when the batch starts).  Destinations are disjoint and in stream
// order, so the matches j depends on are a range of lanes [i_lo, i_hi], found by two binary searches; ranges wider than
// two lanes use the running maximum up to i_hi (never too small: a larger level is only later, not wrong).
// Record:  w0 = M,  w1 = biased destination (mo + rb, rb = out & 15),  w2 = sub-batch (0..63) | level << 16 | class << 24,  w3 = off.
// Sub-batches: consecutive sequences of the batch whose output spans at most ring / 8 bytes (greedy).
// Classes: 0 no match | 1: 4..7 bytes | 2: 8..16 | 3: 17..32 | 4: 33..64 (all not overlapping, source inside the ring) |
//   5 overlapping, <= 64 | 6 whole wave (longer, or wrapping around the ring) | 7 source (partly) older than what the ring
//   is guaranteed to hold when the sub-batch is resolved: moved by the stager | 8 a sequence larger than a sub-batch.
__global__ __launch_bounds__(64) void lzf_seg_records_kernel(seg_ctx c) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[kTileStage];
    __shared__ uint16_t list[kTileTokMax + 64u];
    const uint32_t j = (c.rec_by_len || c.grouped) ? seg_job_of(c, blockIdx.y) : blockIdx.y;
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
    const uint32_t R = c.ring_bytes, kSpan = R / 8u, kAheadB = 3u * kSpan + 64u;
    const uint32_t rb = (uint32_t)(reinterpret_cast<uintptr_t>(job.out) & 15u);
    for (uint32_t t = blockIdx.x; t < sj.ntile; t += gridDim.x) {
        __syncthreads();
        const uint32_t n = tile_enumerate(c, j, t, lane, len, in, stage, list);
        TileCtx tc{t * kSegTile, lds_addr(stage), len, in};
        const uint32_t tbase = c.tile_tok[(size_t)j * c.maxtile + t] * 64u;         // first record of the tile (its batches are padded to 64)
        uint32_t obase = c.tile_out[(size_t)j * c.maxtile + t];
        bool bad = false;
        for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
            const uint32_t nb = n - i0 < 64u ? n - i0 : 64u;
            const bool act = lane < nb;
            Tok k; k.L = 0; k.M = 0; k.off = 0; k.src = 0; k.err = false;
            if (act) k = tile_decode(tc, tc.tstart + list[i0 + lane]);
            const uint32_t tot = k.L + k.M;
            const uint32_t incl = wave_scan_add(tot);
            const uint32_t ob = obase;
            const uint32_t lo = obase + (incl - tot), mo = lo + k.L;
            obase += __builtin_amdgcn_readlane(incl, 63);
            bool ok = act && !k.err;
            if (ok) {
                if (lo > cap || cap - lo < k.L) ok = false;                                   // our buffer (literals)
                else if (k.M && ((uint64_t)mo + k.M > limit || k.off == 0u || k.off > mo || cap - mo < k.M)) ok = false;   // :72-74, :83, :84-89, our buffer
            }
            bad = bad || (act && !ok);
            if (ok) {
                if (k.L > 0u && k.L <= 64u) copy_small_gg(out + lo, in + k.src, k.L);       // literals :65-67
                else if (k.L > 64u && k.L <= 256u) copy_medium_gg(out + lo, in + k.src, k.L);
            }
            for (unsigned long long m = __ballot(ok && k.L > 256u); m; m &= m - 1ull) {       // long runs: all lanes
                const uint32_t q = (uint32_t)__builtin_ctzll(m);
                wave_copy_long(out + __builtin_amdgcn_readlane(lo, q), in + __builtin_amdgcn_readlane(k.src, q), __builtin_amdgcn_readlane(k.L, q), lane);
            }
            if (__ballot(bad)) continue;                                                      // (the job is not ours any more: no records needed)
            // ---- levels: which matches of the batch copy from which (see the head of this section)
            const uint32_t M = k.M, off = k.off;
            const bool has = act && M != 0u;
            const uint32_t span = M < off ? M : off;
            const uint32_t s0 = mo - off, e0 = s0 + span;
            uint32_t lvl = has ? 1u : 0u;
            const bool dep = has && e0 > ob;             // the source reaches into the batch
            // (lane values are exchanged by ds_bpermute: no LDS memory, no address arithmetic — every lane is active here)
            auto from_lane = [](uint32_t l, uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((l & 63u) << 2), (int)v); };
            if (__any(dep)) {
                const uint32_t endv = act ? mo + M : 0xFFFFFFFFu, mov = act ? mo : 0xFFFFFFFFu;
                uint32_t a = 0, bb = 0;                   // a = #lanes with end <= s0; bb = #lanes with mo < e0
#pragma unroll
                for (uint32_t step = 32u; step; step >>= 1) {
                    if (from_lane(a + step - 1u, endv) <= s0) a += step;
                    if (from_lane(bb + step - 1u, mov) < e0) bb += step;
                }
                const uint32_t ilo = a, ihi = bb - 1u;
                const bool any_dep = dep && bb >= 1u && ilo <= ihi && ihi < lane;
                if (__any(any_dep)) {
                    const bool wide = any_dep && ihi - ilo > 1u;
                    const bool any_wide = __any(wide);
                    for (uint32_t it = 0; it < 64u; ++it) {
                        const uint32_t la = from_lane(ilo, lvl), lb = from_lane(ihi, lvl);
                        uint32_t m = la > lb ? la : lb;
                        if (any_wide) { const uint32_t pm = from_lane(ihi, wave_scan_max(lvl)); if (wide) m = pm; }
                        const uint32_t nl = any_dep ? 1u + m : lvl;
                        const bool ch = nl != lvl;
                        lvl = nl;
                        if (!__any(ch)) break;
                    }
                }
            }
            // ---- sub-batches and classes
            const uint32_t endp = mo + M;
            uint32_t sub = 0, cls = 0;
            uint32_t a = 0, sidx = 0;
            while (a < nb) {
                const uint32_t sob = __builtin_amdgcn_readlane(lo, a);
                const uint32_t bb0 = first_lane(__ballot(lane >= a && lane < nb && endp - sob > kSpan));
                uint32_t e = bb0 < nb ? bb0 : nb;
                const bool giant = e == a;
                if (giant) e = a + 1u;
                const uint32_t oe = __builtin_amdgcn_readlane(endp, e - 1u);
                if (lane >= a && lane < e) {
                    sub = sidx;
                    if (giant) cls = 8u;
                    else if (M != 0u) {
                        // what the ring holds for certain when this sub-batch is resolved: from fill pointer + fetch-ahead - ring on
                        const uint32_t fpw = ((oe + rb + 15u) & ~15u) + kAheadB;
                        const uint32_t lov = fpw > R ? fpw - R : 0u;
                        const uint32_t sy = s0 + rb, dy = mo + rb;
                        const uint32_t di = dy & (R - 1u), si = sy & (R - 1u);
                        const bool wrap = di + M > R || si + M > R;
                        if (sy < lov) cls = 7u;
                        else if (M > 64u || wrap) cls = 6u;
                        else if (off < M) cls = (off == 1u || off == 2u || off == 4u) ? (M < 8u ? 9u : M <= 16u ? 10u : M <= 32u ? 11u : 12u) : 5u;
                        else cls = M < 8u ? 1u : M <= 16u ? 2u : M <= 32u ? 3u : 4u;
                    }
                }
                a = e; ++sidx;
            }
            // ---- the records of the batch; the lanes behind the tile's last token pad it (an empty sequence at the batch's end)
            {
                const uint32_t last_end = __builtin_amdgcn_readlane(endp, (nb - 1u) & 63u);
                const uint32_t last_sub = __builtin_amdgcn_readlane(sub, (nb - 1u) & 63u);
                u32x4 w;
                w[0] = act ? M : 0u; w[1] = (act ? mo : last_end) + rb;
                // bits 8..15: the resolver's round masks as flags (one v_and + v_cmp each there): 1 four-byte pieces (classes 1, 9),
                // 2 first pair of eight-byte pieces (2-4, 10-12), 4 second pair (3, 4, 11, 12), 8 third and fourth (4, 12),
                // 16 leaves the loop (5, 6), 32 run-length (9-12)
                const uint32_t fl = !act ? 0u : ((cls == 1u || cls == 9u) ? 1u : 0u) | (((cls >= 2u && cls <= 4u) || (cls >= 10u && cls <= 12u)) ? 2u : 0u) |
                                    ((cls == 3u || cls == 4u || cls == 11u || cls == 12u) ? 4u : 0u) | ((cls == 4u || cls == 12u) ? 8u : 0u) |
                                    ((cls == 5u || cls == 6u) ? 16u : 0u) | ((cls >= 9u && cls <= 12u) ? 32u : 0u);
                w[2] = (act ? sub : last_sub) | (fl << 8) | ((act ? lvl : 0u) << 16) | ((act ? cls : 0u) << 24); w[3] = act ? off : 0u;
                recs[tbase + i0 + lane] = w;
            }
        }
        if (__ballot(bad) && lane == 0u) c.st[j].failed = 1u;
    }
}

// =====================================================================================================================
// order of the resolve stage's workgroups (kernels.h): one workgroup, n_jobs <= 1024
// =====================================================================================================================
__global__ __launch_bounds__(1024) void lzf_seg_by_len_kernel(seg_ctx c) {
    __shared__ uint32_t cost[1024];
    const uint32_t i = threadIdx.x, n = c.n_jobs;
    if (n > 1024u) return;                                        // (one workgroup ranks the call: the dispatch never asks for more, capi.hip kSegRankMax)
    if (i < n) { const uint64_t l = c.jobs[i].input_len; cost[i] = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l; }
    __syncthreads();
    if (i >= n) return;
    const uint32_t mine = cost[i];
    uint32_t r = 0;
    for (uint32_t k = 0; k < n; ++k) { const uint32_t o = cost[k]; r += (o > mine || (o == mine && k < i)) ? 1u : 0u; }
    c.by_len[r] = i;
}
// Grouped calls: one wavefront that does nothing for `ticks` of the device's wall clock (capi.hip converts from microseconds by the rate the runtime reports).  It sits on the caller's stream between a group's
// records stage and the next one's: the group's resolve stage starts on another stream (an event away: a few microseconds later), and
// its workgroups — 32+ KiB of LDS each — should find the compute units empty rather than squeeze in between the next records stage's
// (measured at 980 blocks: 15.4 ms with such a pause, 17.0 without).
__global__ __launch_bounds__(64) void lzf_seg_pause_kernel(uint32_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
// Grouped calls: the jobs of the call by their sequences, most first — by_tok[rank] = job; consecutive ranks form the groups that go
// through the last two stages together (sizes gs.x .. gs.w, zero = no such group) — and, where the resolve stage orders its
// workgroups (c.order), every group's order at once: a group's jobs are ranks off .. off + size - 1, dealt out in rows of n_cu
// like lzf_seg_order_kernel does for a whole call.
__global__ __launch_bounds__(1024) void lzf_seg_rank_kernel(seg_ctx c, uint32_t* __restrict__ by_tok, uint4 gs) {
    __shared__ uint32_t cost[1024];
    const uint32_t i = threadIdx.x, n = c.n_jobs;
    if (n > 1024u) return;
    if (i < n) { const seg_job s = c.st[i]; cost[i] = (s.eligible && !s.failed) ? s.ntok : 0u; }
    __syncthreads();
    if (i >= n) return;
    const uint32_t mine = cost[i];
    uint32_t r = 0;
    for (uint32_t k = 0; k < n; ++k) { const uint32_t o = cost[k]; r += (o > mine || (o == mine && k < i)) ? 1u : 0u; }
    by_tok[r] = i;
    if (!c.order) return;
    const uint32_t sz[4] = {gs.x, gs.y, gs.z, gs.w};
    uint32_t off = 0, gn = 0;
    for (uint32_t k = 0; k < 4u; ++k) { if (r >= off + sz[k]) off += sz[k]; else { gn = sz[k]; break; } }
    if (!gn) return;
    const uint32_t rr = r - off;
    const uint32_t ncu = c.n_cu ? c.n_cu : 256u;
    const uint32_t row = rr / ncu, col = rr % ncu;
    const uint32_t rowlen = gn - row * ncu < ncu ? gn - row * ncu : ncu;
    c.order[off + row * ncu + ((row & 1u) ? rowlen - 1u - col : col)] = i;
}
__global__ __launch_bounds__(1024) void lzf_seg_order_kernel(seg_ctx c) {
    __shared__ uint32_t cost[1024];
    const uint32_t i = threadIdx.x, n = c.g_n;                     // (the ranks of this launch's group)
    if (n > 1024u) return;
    const uint32_t job = i < n ? (c.grouped ? seg_job_of(c, i) : i) : 0u;
    if (i < n) { const seg_job s = c.st[job]; cost[i] = (s.eligible && !s.failed) ? s.ntok : 0u; }
    __syncthreads();
    if (i >= n) return;
    const uint32_t mine = cost[i];
    uint32_t r = 0;
    for (uint32_t k = 0; k < n; ++k) { const uint32_t o = cost[k]; r += (o > mine || (o == mine && k < i)) ? 1u : 0u; }
    const uint32_t ncu = c.n_cu ? c.n_cu : 256u;
    const uint32_t row = r / ncu, col = r % ncu;
    const uint32_t rowlen = n - row * ncu < ncu ? n - row * ncu : ncu;
    c.order[c.g_off + row * ncu + ((row & 1u) ? rowlen - 1u - col : col)] = job;
}

// =====================================================================================================================
// resolve: the dependent match copies of a block — a pair of wavefronts per block
// =====================================================================================================================
// Biased positions y = x + rb (rb = out & 15): y % 16 == 0 <=> out + x is 16-byte aligned; ring index = y & (R - 1).
// The STAGER (wave 1) does everything that is not the dependency chain: the ring's granules in from `out` (the literals are
// there), finished granules out to HBM, the few sources that are older than the ring (class 7: from HBM), sequences larger
// than a sub-batch (class 8: HBM -> HBM, then the ring's history is read back).  The RESOLVER (wave 0) runs the rounds of
// sub-batch after sub-batch: one LDS round trip per dependency level.  Both read the records (records stage) from HBM on
// their own, two batches ahead of their use.
// Tickets = sub-batches in stream order: the stager publishes ticket t with ctl[0] = t + 1 after its LDS writes, the
// resolver answers ctl[1] = t + 1 after its own; LDS executes one wave's accesses in order, so a flag is never seen before
// the data.  The stager runs at most NT tickets and NS sub-batch spans of output ahead (and loads the granules behind its fill pointer one sub-batch early);
// the records stage classed the sources with that fetch-ahead in mind.
template <int R>
__global__ __launch_bounds__(128) void lzf_seg_resolve_pair_kernel(seg_ctx c) {
    constexpr uint32_t kMask = (uint32_t)R - 1u;
    constexpr uint32_t kSpan = (uint32_t)R / 8u;       // output bytes one sub-batch may produce
    constexpr uint32_t NS = 3;                         // sub-batch spans of output the stager may be ahead (what the records stage classed the sources for)
    constexpr uint32_t NT = 32;                        // tickets it may be ahead (a text block's sub-batch is a batch, ~1 KiB: three were 6 us of slack, less than HBM answers in under load)
    // (the ring starts 64 bytes into the workgroup's LDS: an address 32 bytes in front of a ring position is never negative)
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[64 + R];
    uint8_t* const ring = lds_all + 64;
    uint32_t* const ctl = reinterpret_cast<uint32_t*>(lds_all);      // [0] staged tickets, [1] resolved tickets, [3] a wave gave up
    if (blockIdx.x >= c.g_n) return;
    const uint32_t j = c.order ? c.order[c.g_off + blockIdx.x] : c.grouped ? seg_job_of(c, blockIdx.x) : blockIdx.x;
    const seg_job sj = c.st[j];
    if (!sj.eligible || sj.failed) return;
    if (c.ring_bytes != (uint32_t)R) return;           // (the records were classed for another ring: leave the job to the pair kernel)
    if (c.res_prio) __builtin_amdgcn_s_setprio(3);     // the block's chain: ahead of the throughput kernels that share the SIMD
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
    if (threadIdx.x < 4u) ctl[threadIdx.x] = 0u;
    __syncthreads();
    // the flags, by explicit DS instructions (a volatile pointer to ctl[] is a generic pointer to hipcc: FLAT accesses, which
    // wait for every global load in flight as well)
    const uint32_t ctl_a = lds_addr(ctl);
    auto flag_get = [&](uint32_t i) -> uint32_t { uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ctl_a + 4u * i) : "memory"); return __builtin_amdgcn_readfirstlane(v); };
    auto flag_set = [&](uint32_t i, uint32_t v) { asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" ::"v"(ctl_a + 4u * i), "v"(v) : "memory"); };
    // Wait until flag i exceeds `below`.  Bounded: a wait of ~2^22 naps (seconds; a ticket takes microseconds) can only mean a
    // defect, so the wave gives up and raises ctl[3]; both waves then leave, the job is not marked done and the pair kernel
    // (launched after this one) decodes it.  false = give up.
    auto flag_wait_above = [&](uint32_t i, uint32_t below) -> bool {
        for (uint32_t spin = 0;; ++spin) {
            if (flag_get(i) > below) return true;
            if ((spin & 63u) == 63u && (flag_get(3) != 0u || spin > (1u << 22))) { if (lane == 0u) flag_set(3, 1u); return false; }
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // ring[d, d + nbytes) <- ring[s, s + nbytes): source final, ranges disjoint; all lanes, 8 bytes each, the last unit moved
    // back to end with the range (it overlaps its neighbour with the same bytes): one LDS round trip per 512 bytes
    auto ring_copy = [&](uint32_t d, uint32_t s_, uint32_t nbytes) {
        if (nbytes >= 8u) {
            const uint32_t nu = (nbytes + 7u) >> 3;
            for (uint32_t u = lane; u < nu; u += kWave) {
                const uint32_t o = 8u * u + 8u <= nbytes ? 8u * u : nbytes - 8u;
                const uint32_t sa = (s_ + o) & kMask, da = (d + o) & kMask;
                if (sa + 8u <= (uint32_t)R && da + 8u <= (uint32_t)R) lds_st64(ring_a + da, lds_ld64u(ring_a + sa));
                else for (uint32_t t = 0; t < 8u; ++t) ring[(da + t) & kMask] = ring[(sa + t) & kMask];
            }
        } else if (lane < nbytes) ring[(d + lane) & kMask] = ring[(s_ + lane) & kMask];
    };
    // copy_overlapping (decompress.rs:80-138) inside the ring: out[d + t] = out[d - off + t], t < M.  The bytes already
    // copied double the source every step (a multiple of the period is a period).
    auto ring_match = [&](uint32_t d, uint32_t off_, uint32_t M_) {
        uint32_t pos = 0, av = off_;
        while (pos < M_) { const uint32_t cc = av < M_ - pos ? av : M_ - pos; ring_copy(d + pos, d - off_, cc); pos += cc; av += cc; }
    };

    if (role == 0u) {
        // ================================================ RESOLVER ================================================
#ifdef LZF_SEG_TIME
        long long tm_wait = 0, tm_setup = 0, tm_asm = 0, tm_slow = 0, tm_t = clock64(); uint32_t n_rounds = 0;
#define RT(var) do { const long long t__ = clock64(); var += t__ - tm_t; tm_t = t__; } while (0)
#else
#define RT(var) do { } while (0)
#endif
        uint32_t t = 0, staged = 0;                       // tickets resolved; tickets known to be staged (the flag as last read)
        // the records come straight from HBM, two batches ahead (the only loads of this wave: they return in order and in time)
        u32x4 rA = u32x4{0, 0, 0, 0}, rB = u32x4{0, 0, 0, 0};
        if (n) { rA = recs[lane < n ? lane : n - 1u]; rB = recs[64u + lane < n ? 64u + lane : n - 1u]; }
        auto wait_staged = [&](uint32_t tt) -> bool {    // ticket tt published?  (one read of the flag usually covers several tickets)
            if (staged > tt) return true;
            for (uint32_t spin = 0;; ++spin) {
                staged = flag_get(0);
                if (staged > tt) return true;
                if ((spin & 63u) == 63u && (flag_get(3) != 0u || spin > (1u << 22))) { if (lane == 0u) flag_set(3, 1u); return false; }
                __builtin_amdgcn_s_sleep(1);
            }
        };
        for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
            const uint32_t nb = n - i0 < 64u ? n - i0 : 64u;
            const u32x4 r = rA;
            asm volatile("" ::: "memory");
            rA = rB;
            { const uint32_t ix = i0 + 128u + lane; rB = recs[ix < n ? ix : n - 1u]; }
            asm volatile("" ::: "memory");
            RT(tm_setup);
            if (!wait_staged(t)) break;
            RT(tm_wait);
            const uint32_t M = r[0], dy = r[1], off = r[3];
            const uint32_t sub = lane < nb ? (r[2] & 0xFFu) : 0xFFu, lvl = (r[2] >> 16) & 255u, cls = lane < nb ? r[2] >> 24 : 0u;
            const uint32_t nsub = __builtin_amdgcn_readlane(r[2] & 0xFFu, (nb - 1u) & 63u) + 1u;
            const uint32_t sa = ring_a + ((dy - off) & kMask), da = ring_a + (dy & kMask);
            // two-ended pieces: A (4..7) = 4-byte pieces at 0 and M - 4; B1 (8..16) = 8-byte pieces at 0 and M - 8;
            // B2 (17..32) adds 8 and M - 16; C (33..64) adds 16, 24, M - 32, M - 24
            const uint32_t o3 = M - 8u, a1 = M - 4u;
#if !defined(LZF_SEG_NOASM) && !defined(LZF_SEG_DBG_SKIP) && !defined(LZF_SEG_NORLE)
            // classes 9..12: run-length matches (offset 1, 2 or 4) of the same four sizes — stored like 1..4, the registers filled
            // with the pattern (the first bytes read, spread by v_perm_b32; rotated for the piece that ends the match)
            const uint32_t fl = r[2] >> 8;              // (the records stage's flags; padding lanes carry none)
            const unsigned long long mA_b = __ballot((fl & 1u) != 0u), m2_b = __ballot((fl & 2u) != 0u), m3_b = __ballot((fl & 4u) != 0u),
                                     m4_b = __ballot((fl & 8u) != 0u), mS_b = __ballot((fl & 16u) != 0u), mR_b = __ballot((fl & 32u) != 0u);
            uint32_t rsel = 0, rsh = 0;                  // (only batches with run-length lanes pay for these)
            if (mR_b) { rsel = off == 1u ? 0u : off == 2u ? 0x01000100u : 0x03020100u; rsh = M & (off - 1u); }
#else
            // (the compiler's loop: the run-length classes take the general path (a) below)
            const unsigned long long mA_b = __ballot(cls == 1u), m2_b = __ballot(cls >= 2u && cls <= 4u), m3_b = __ballot(cls == 3u || cls == 4u),
                                     m4_b = __ballot(cls == 4u), mS_b = __ballot(cls == 5u || cls == 6u || (cls >= 9u && cls <= 12u));
#ifdef LZF_SEG_NORLE
            const unsigned long long mR_b = 0; const uint32_t rsel = 0, rsh = 0;
#endif
#endif
            bool gave_up = false;
            for (uint32_t s_i = 0; s_i < nsub; ++s_i, ++t) {
#ifdef LZF_ANALYSIS      // LZF_SEG_FORCE=resolver: the resolver of every odd job gives up at its third ticket (test of the hand-over)
                if (c.dbg_force == 2u && (j & 1u) && t >= 2u) { if (lane == 0u) flag_set(3, 1u); gave_up = true; break; }
#endif
                // ---- wait for the stager's ticket
                if (s_i) {
                    RT(tm_setup);
                    if (!wait_staged(t)) { gave_up = true; break; }
                    RT(tm_wait);
                }
                const unsigned long long msub = nsub == 1u ? ~0ull : __ballot(sub == s_i);
                unsigned long long todo = (mA_b | m2_b | mS_b) & msub;
#if defined(LZF_SEG_DBG_SKIP) && LZF_SEG_DBG_SKIP == 2      // analysis: no rounds at all (what a batch costs without them)
                todo = 0;
#endif
                for (uint32_t lv = 1; todo; ++lv) {
                    unsigned long long ml;
#if !defined(LZF_SEG_NOASM) && !defined(LZF_SEG_DBG_SKIP)
                    // The rounds of the two-ended classes in one hand-scheduled loop: a lone wavefront retires an instruction every
                    // 5 to 8 cycles, so a round is priced by its instruction count (about 20 scalar instructions + the DS pairs of
                    // the classes present; what hipcc makes of the loop below is 45 + 8 EXEC moves).  It returns after a level
                    // that has lanes of class 5/6 (`ts`; the level's other lanes done) for the code below.  v100..v117: the
                    // pieces in flight, fixed registers so that the halves of a pair can be named.
                    unsigned long long ts;
                    {
                        unsigned long long sv;
                        lv = __builtin_amdgcn_readfirstlane(lv);
                        // (two copies: batches without run-length and class-5/6 lanes — most of a text block's — skip those checks)
                        if ((mR_b | mS_b) & msub)
                        asm volatile(
                            "s_mov_b64 %[sv], exec\n\t"
                            ".p2align 6\n\t"                                     // (the loop starts an instruction-cache line: its time must not depend on what is compiled around it)
                            "Lloop%=:\n\t"
                            "v_cmp_eq_u32_e32 vcc, %[lv], %[vl]\n\t"
                            "s_add_u32 %[lv], %[lv], 1\n\t"
                            "s_and_b64 %[ml], vcc, %[todo]\n\t"
                            "s_cbranch_scc0 Lempty%=\n\t"
#ifdef LZF_SEG_TIME
                            "s_add_u32 %[nr], %[nr], 1\n\t"
#endif
                            "s_and_b64 exec, %[ml], %[mA]\n\t"
                            "ds_read_b32 v116, %[sa]\n\t"
                            "ds_read_b32 v117, %[tsrc] offset:28\n\t"
                            "s_and_b64 exec, %[ml], %[m2]\n\t"
                            "ds_read_b64 v[100:101], %[sa]\n\t"
                            "ds_read_b64 v[106:107], %[tsrc] offset:24\n\t"
                            "s_and_b64 exec, %[ml], %[m3]\n\t"
                            "s_cbranch_execz Lr%=\n\t"
                            "ds_read_b64 v[102:103], %[sa] offset:8\n\t"
                            "ds_read_b64 v[104:105], %[tsrc] offset:16\n\t"
                            "s_and_b64 exec, %[ml], %[m4]\n\t"
                            "s_cbranch_execz Lr%=\n\t"
                            "ds_read_b64 v[108:109], %[sa] offset:16\n\t"
                            "ds_read_b64 v[110:111], %[sa] offset:24\n\t"
                            "ds_read_b64 v[112:113], %[tsrc]\n\t"
                            "ds_read_b64 v[114:115], %[tsrc] offset:8\n\t"
                            "Lr%=:\n\t"
                            "s_and_b64 exec, %[ml], %[mR]\n\t"
#ifndef LZF_SEG_DBG_NOWAIT                                         // (analysis: the round without its LDS latency; wrong bytes)
                            "s_waitcnt lgkmcnt(0)\n\t"
#endif
                            "s_cbranch_execz Lnr%=\n\t"
                            // run-length lanes: the pattern (what was read at the source, spread over the word) in every piece
                            // from the front, the pattern rotated by M mod offset in every piece that ends with the match
                            "v_perm_b32 v116, v116, v116, %[rsel]\n\t"
                            "v_perm_b32 v100, v100, v100, %[rsel]\n\t"
                            "v_alignbyte_b32 v117, v116, v116, %[rsh]\n\t"
                            "v_alignbyte_b32 v106, v100, v100, %[rsh]\n\t"
                            "v_mov_b32 v101, v100\n\t"
                            "v_mov_b32 v107, v106\n\t"
                            "v_mov_b64 v[102:103], v[100:101]\n\t"
                            "v_mov_b64 v[104:105], v[106:107]\n\t"
                            "v_mov_b64 v[108:109], v[100:101]\n\t"
                            "v_mov_b64 v[110:111], v[100:101]\n\t"
                            "v_mov_b64 v[112:113], v[106:107]\n\t"
                            "v_mov_b64 v[114:115], v[106:107]\n\t"
                            "Lnr%=:\n\t"
                            "s_and_b64 exec, %[ml], %[mA]\n\t"
                            "ds_write_b32 %[da], v116\n\t"
                            "ds_write_b32 %[tdst], v117 offset:28\n\t"
                            "s_and_b64 exec, %[ml], %[m2]\n\t"
                            "ds_write_b64 %[da], v[100:101]\n\t"
                            "ds_write_b64 %[tdst], v[106:107] offset:24\n\t"
                            "s_and_b64 exec, %[ml], %[m3]\n\t"
                            "s_cbranch_execz Lw%=\n\t"
                            "ds_write_b64 %[da], v[102:103] offset:8\n\t"
                            "ds_write_b64 %[tdst], v[104:105] offset:16\n\t"
                            "s_and_b64 exec, %[ml], %[m4]\n\t"
                            "s_cbranch_execz Lw%=\n\t"
                            "ds_write_b64 %[da], v[108:109] offset:16\n\t"
                            "ds_write_b64 %[da], v[110:111] offset:24\n\t"
                            "ds_write_b64 %[tdst], v[112:113]\n\t"
                            "ds_write_b64 %[tdst], v[114:115] offset:8\n\t"
                            "Lw%=:\n\t"
                            "s_mov_b64 exec, %[sv]\n\t"
                            "s_and_b64 %[ts], %[ml], %[mS]\n\t"
                            "s_cbranch_scc1 Lout%=\n\t"
                            "s_andn2_b64 %[todo], %[todo], %[ml]\n\t"
                            "s_cbranch_scc1 Lloop%=\n\t"
                            "s_branch Lout%=\n\t"
                            "Lempty%=:\n\t"
                            "s_mov_b64 %[ts], 0\n\t"
                            "s_cmp_le_u32 %[lv], 71\n\t"
                            "s_cbranch_scc1 Lloop%=\n\t"
                            "Lout%=:\n\t"
                            : [sv] "=&s"(sv), [ts] "=&s"(ts), [ml] "=&s"(ml), [todo] "+s"(todo), [lv] "+s"(lv)
#ifdef LZF_SEG_TIME
                              , [nr] "+s"(n_rounds)
#endif
                            : [mA] "s"(mA_b), [m2] "s"(m2_b), [m3] "s"(m3_b), [m4] "s"(m4_b), [mS] "s"(mS_b), [mR] "s"(mR_b), [vl] "v"(lvl), [rsel] "v"(rsel), [rsh] "v"(rsh),
                              [sa] "v"(sa), [tsrc] "v"(sa + M - 32u),
                              [da] "v"(da), [tdst] "v"(da + M - 32u)
                            : "memory", "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",
                              "v114", "v115", "v116", "v117");
                        else
                        asm volatile(
                            "s_mov_b64 %[sv], exec\n\t"
                            "s_mov_b64 %[ts], 0\n\t"
                            ".p2align 6\n\t"                                     // (the loop starts an instruction-cache line: its time must not depend on what is compiled around it)
                            "Lloop%=:\n\t"
                            "v_cmp_eq_u32_e32 vcc, %[lv], %[vl]\n\t"
                            "s_add_u32 %[lv], %[lv], 1\n\t"
                            "s_and_b64 %[ml], vcc, %[todo]\n\t"
                            "s_cbranch_scc0 Lempty%=\n\t"
#ifdef LZF_SEG_TIME
                            "s_add_u32 %[nr], %[nr], 1\n\t"
#endif
                            "s_and_b64 exec, %[ml], %[mA]\n\t"
                            "ds_read_b32 v116, %[sa]\n\t"
                            "ds_read_b32 v117, %[tsrc] offset:28\n\t"
                            "s_and_b64 exec, %[ml], %[m2]\n\t"
                            "ds_read_b64 v[100:101], %[sa]\n\t"
                            "ds_read_b64 v[106:107], %[tsrc] offset:24\n\t"
                            "s_and_b64 exec, %[ml], %[m3]\n\t"
                            "s_cbranch_execz Lr%=\n\t"
                            "ds_read_b64 v[102:103], %[sa] offset:8\n\t"
                            "ds_read_b64 v[104:105], %[tsrc] offset:16\n\t"
                            "s_and_b64 exec, %[ml], %[m4]\n\t"
                            "s_cbranch_execz Lr%=\n\t"
                            "ds_read_b64 v[108:109], %[sa] offset:16\n\t"
                            "ds_read_b64 v[110:111], %[sa] offset:24\n\t"
                            "ds_read_b64 v[112:113], %[tsrc]\n\t"
                            "ds_read_b64 v[114:115], %[tsrc] offset:8\n\t"
                            "Lr%=:\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_and_b64 exec, %[ml], %[mA]\n\t"
                            "ds_write_b32 %[da], v116\n\t"
                            "ds_write_b32 %[tdst], v117 offset:28\n\t"
                            "s_and_b64 exec, %[ml], %[m2]\n\t"
                            "ds_write_b64 %[da], v[100:101]\n\t"
                            "ds_write_b64 %[tdst], v[106:107] offset:24\n\t"
                            "s_and_b64 exec, %[ml], %[m3]\n\t"
                            "s_cbranch_execz Lw%=\n\t"
                            "ds_write_b64 %[da], v[102:103] offset:8\n\t"
                            "ds_write_b64 %[tdst], v[104:105] offset:16\n\t"
                            "s_and_b64 exec, %[ml], %[m4]\n\t"
                            "s_cbranch_execz Lw%=\n\t"
                            "ds_write_b64 %[da], v[108:109] offset:16\n\t"
                            "ds_write_b64 %[da], v[110:111] offset:24\n\t"
                            "ds_write_b64 %[tdst], v[112:113]\n\t"
                            "ds_write_b64 %[tdst], v[114:115] offset:8\n\t"
                            "Lw%=:\n\t"
                            "s_mov_b64 exec, %[sv]\n\t"
                            "s_andn2_b64 %[todo], %[todo], %[ml]\n\t"
                            "s_cbranch_scc1 Lloop%=\n\t"
                            "s_branch Lout%=\n\t"
                            "Lempty%=:\n\t"
                            "s_mov_b64 %[ts], 0\n\t"
                            "s_cmp_le_u32 %[lv], 71\n\t"
                            "s_cbranch_scc1 Lloop%=\n\t"
                            "Lout%=:\n\t"
                            : [sv] "=&s"(sv), [ts] "=&s"(ts), [ml] "=&s"(ml), [todo] "+s"(todo), [lv] "+s"(lv)
#ifdef LZF_SEG_TIME
                              , [nr] "+s"(n_rounds)
#endif
                            : [mA] "s"(mA_b), [m2] "s"(m2_b), [m3] "s"(m3_b), [m4] "s"(m4_b), [mS] "s"(mS_b), [mR] "s"(mR_b), [vl] "v"(lvl), [rsel] "v"(rsel), [rsh] "v"(rsh),
                              [sa] "v"(sa), [tsrc] "v"(sa + M - 32u),
                              [da] "v"(da), [tdst] "v"(da + M - 32u)
                            : "memory", "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",
                              "v114", "v115", "v116", "v117");
                        RT(tm_asm);
                        // (hipcc takes what an asm statement returns for divergent, whatever the register class)
                        auto uni = [](unsigned long long x) { return (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)x) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x >> 32)) << 32); };
                        ts = uni(ts); ml = uni(ml); todo = uni(todo); lv = __builtin_amdgcn_readfirstlane(lv);
                        if (!ts) break;                  // nothing left (or levels that do not end: not our records)
                        todo &= ~ml;                     // (a level left for its class-5/6 lanes is still in `todo`)
                        --lv;                            // (the loop's increment follows)
                    }
#else
                    ml = __ballot(lvl == lv) & todo;
                    if (!ml) { if (lv > 70u) break; continue; }
                    todo &= ~ml;
#endif
#if !defined(LZF_SEG_NOASM) && !defined(LZF_SEG_DBG_SKIP)
                    const unsigned long long mA = 0, m2 = 0, m3 = 0, m4 = 0, mS = ts;
#else
#ifdef LZF_SEG_TIME
                    ++n_rounds;
#endif
                    const unsigned long long mA = ml & mA_b, m2 = ml & m2_b, m3 = ml & m3_b, m4 = ml & m4_b, mS = ml & mS_b;
#endif
                    RT(tm_setup);
#if defined(LZF_SEG_DBG_SKIP) && LZF_SEG_DBG_SKIP == 1      // analysis: the level loop without its copies (what the control around a round costs)
                    continue;
#endif
                    if (mA | m2) {
                        uint64_t v0, v1, v2, v3, v4, v5, v6, v7; uint32_t va0, va1; unsigned long long sv;
                        asm volatile(
                            "s_mov_b64 %[sv], exec\n\t"
                            "s_mov_b64 exec, %[mA]\n\t"
                            "ds_read_b32 %[va0], %[sa]\n\t"
                            "ds_read_b32 %[va1], %[s1]\n\t"
                            "s_mov_b64 exec, %[m2]\n\t"
                            "ds_read_b64 %[v0], %[sa]\n\t"
                            "ds_read_b64 %[v3], %[s3]\n\t"
                            "s_mov_b64 exec, %[m3]\n\t"
                            "s_cbranch_execz Lr3%=\n\t"
                            "ds_read_b64 %[v1], %[sa] offset:8\n\t"
                            "ds_read_b64 %[v2], %[s16]\n\t"
                            "s_mov_b64 exec, %[m4]\n\t"
                            "s_cbranch_execz Lr3%=\n\t"
                            "ds_read_b64 %[v4], %[sa] offset:16\n\t"
                            "ds_read_b64 %[v5], %[sa] offset:24\n\t"
                            "ds_read_b64 %[v6], %[s32]\n\t"
                            "ds_read_b64 %[v7], %[s32] offset:8\n\t"
                            "Lr3%=:\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_mov_b64 exec, %[mA]\n\t"
                            "ds_write_b32 %[da], %[va0]\n\t"
                            "ds_write_b32 %[d1], %[va1]\n\t"
                            "s_mov_b64 exec, %[m2]\n\t"
                            "ds_write_b64 %[da], %[v0]\n\t"
                            "ds_write_b64 %[d3], %[v3]\n\t"
                            "s_mov_b64 exec, %[m3]\n\t"
                            "s_cbranch_execz Lw3%=\n\t"
                            "ds_write_b64 %[da], %[v1] offset:8\n\t"
                            "ds_write_b64 %[d16], %[v2]\n\t"
                            "s_mov_b64 exec, %[m4]\n\t"
                            "s_cbranch_execz Lw3%=\n\t"
                            "ds_write_b64 %[da], %[v4] offset:16\n\t"
                            "ds_write_b64 %[da], %[v5] offset:24\n\t"
                            "ds_write_b64 %[d32], %[v6]\n\t"
                            "ds_write_b64 %[d32], %[v7] offset:8\n\t"
                            "Lw3%=:\n\t"
                            "s_mov_b64 exec, %[sv]\n\t"
                            : [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3), [v4] "=&v"(v4), [v5] "=&v"(v5), [v6] "=&v"(v6), [v7] "=&v"(v7),
                              [va0] "=&v"(va0), [va1] "=&v"(va1), [sv] "=&s"(sv)
                            : [mA] "s"(mA), [m2] "s"(m2), [m3] "s"(m3), [m4] "s"(m4), [sa] "v"(sa), [s1] "v"(sa + a1), [s3] "v"(sa + o3), [s16] "v"(sa + M - 16u), [s32] "v"(sa + M - 32u),
                              [da] "v"(da), [d1] "v"(da + a1), [d3] "v"(da + o3), [d16] "v"(da + M - 16u), [d32] "v"(da + M - 32u)
                            : "memory");
                    }
                    RT(tm_asm);
                    if (mS) {
                        // (this path is rare: its operands pass through an empty asm statement, or hipcc computes its predicates and
                        //  addresses — some forty instructions — in the prologue of every batch)
                        uint32_t M_s = M, off_s = off, dy_s = dy;
                        asm volatile("" : "+v"(M_s), "+v"(off_s), "+v"(dy_s));
                        const uint32_t M = M_s, off = off_s, dy = dy_s;
                        const uint32_t sa = ring_a + ((dy - off) & kMask), da = ring_a + (dy & kMask);
                        const bool nowS = (mS >> lane) & 1ull;
                        const bool fits = nowS && (((dy - off) & kMask) + M <= (uint32_t)R) && ((dy & kMask) + M <= (uint32_t)R);   // neither range wraps
                        // (a) run-length matches (offset 1, 2 or 4): the pattern from one read, stores only.  Up to 64 bytes in
                        // the lane; longer ones by the whole wave, one after the other (512 bytes a step), the lane's own
                        // store finishing the odd end
                        const bool rle = fits && (off == 1u || off == 2u || off == 4u);
                        if (__ballot(rle)) {
                            uint32_t w = 0;
                            if (rle) {
                                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(sa) : "memory");
                                if (off == 1u) w = (w & 0xFFu) * 0x01010101u; else if (off == 2u) w = (w & 0xFFFFu) * 0x00010001u;
                                const uint64_t pat = (uint64_t)w | ((uint64_t)w << 32);
                                if (M >= 8u) {
                                    if (M <= 64u) for (uint32_t k = 0; k + 8u < M; k += 8u) lds_st64(da + k, pat);
                                    // the last piece ends exactly at M: its phase is (M - 8) mod off (off divides 8): rotate by bytes
                                    const uint32_t ph = (M - 8u) & (off - 1u);
                                    const uint64_t rot = ph ? (pat >> (8u * ph)) | (pat << (64u - 8u * ph)) : pat;
                                    lds_st64(da + M - 8u, rot);
                                } else {
                                    lds_st32(da, w);
                                    const uint32_t ph = (M - 4u) & (off - 1u);
                                    lds_st32(da + M - 4u, ph ? (uint32_t)(pat >> (8u * ph)) : w);
                                }
                            }
                            for (unsigned long long m = __ballot(rle && M > 64u); m; m &= m - 1ull) {
                                const uint32_t q = (uint32_t)__builtin_ctzll(m);
                                const uint32_t qd = __builtin_amdgcn_readlane(da, q), qM = __builtin_amdgcn_readlane(M, q);
                                const uint32_t qw = __builtin_amdgcn_readlane(w, q);
                                const uint64_t qp = (uint64_t)qw | ((uint64_t)qw << 32);
                                for (uint32_t k = 8u * lane; k + 8u <= qM; k += 8u * kWave) lds_st64(qd + k, qp);
                            }
                        }
                        // (b) not overlapping, 65..160 bytes, four or more of them in the level: 32 bytes per step in the lane,
                        // the last step two-ended (fewer: the whole wave is quicker, (d))
                        const bool lngc = fits && !rle && off >= M && M <= 160u;
                        const bool lng = lngc && __builtin_popcountll(__ballot(lngc)) >= 4;
                        if (lng) {
                            uint32_t k = 0;
                            for (;;) {
                                const uint32_t kk = k + 32u <= M ? k : M - 32u;
                                uint64_t x0, x1, x2, x3;
                                lds_ld64x4(sa + kk, sa + kk + 8u, sa + kk + 16u, sa + kk + 24u, x0, x1, x2, x3);
                                lds_st64(da + kk, x0); lds_st64(da + kk + 8u, x1); lds_st64(da + kk + 16u, x2); lds_st64(da + kk + 24u, x3);
                                if (k + 32u >= M) break;
                                k += 32u;
                            }
                        }
                        // (c) overlapping, at most 64 bytes: doubling steps of its own
                        const bool dbl = fits && !rle && !lng && M <= 64u;
                        if (dbl) {
                            uint32_t pos = 0, av = off;
                            while (pos < M) { const uint32_t cc = av < M - pos ? av : M - pos; put_small_lds(da + pos, sa, cc); pos += cc; av += cc; }
                        }
                        // (d) the rest, one at a time by the whole wave
                        for (unsigned long long m = __ballot(nowS && !rle && !lng && !dbl); m; m &= m - 1ull) {
                            const uint32_t q = (uint32_t)__builtin_ctzll(m);
                            ring_match(__builtin_amdgcn_readlane(dy, q), __builtin_amdgcn_readlane(off, q), __builtin_amdgcn_readlane(M, q));
                        }
                        RT(tm_slow);
                    }
                }
                if (lane == 0u) flag_set(1, t + 1u);
            }
            if (gave_up) break;
        }
#ifdef LZF_SEG_TIME
        if (lane == 0u) flag_set(2, (uint32_t)(tm_wait >> 10));      // (to the stager, which writes the job's result: results[].reserved = this | total << 16, in units of 2^10 / 2^14 cycles)
        if (lane == 0u) { reinterpret_cast<uint16_t*>(&c.st[j].pad)[0] = (uint16_t)(n_rounds >> 4);
            c.st[j].pad2 = (unsigned long long)(uint16_t)(tm_wait >> 14) | ((unsigned long long)(uint16_t)(tm_setup >> 14) << 16) | ((unsigned long long)(uint16_t)(tm_asm >> 14) << 32) | ((unsigned long long)(uint16_t)(tm_slow >> 14) << 48); }
#endif
#undef RT
    } else {
        // ================================================= STAGER =================================================
        uint32_t fp = 0, fl = rb, safe = 0;              // filled up to (granule), flushed up to, own stores visible below
        uint32_t ticket = 0;                             // tickets published
        static_assert(NT <= 32, "the ends of the tickets in flight live in the lanes of one register");
        uint32_t endv = rb;                              // lane t % 64: biased end (rounded down to a granule) of published ticket t
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
        // the granules behind the fill pointer, loaded one sub-batch early (the load's latency runs under the hand-over)
        u32x4 pfa = u32x4{0, 0, 0, 0}, pfb = u32x4{0, 0, 0, 0};
        uint32_t pf0 = 0, pf1 = 0;                       // [pf0, pf1) is held in registers
        const uint32_t fill_lim = (total + rb + 15u) & ~15u;
        auto prefetch_issue = [&](uint32_t need) {       // (the fill pointer stays within 3 KiB of what the sub-batch needed: the levels' fetch-ahead bound)
            pf0 = fp; pf1 = fp + 2048u < fill_lim ? fp + 2048u : fill_lim;
            if (pf1 <= pf0 || fp > need + 1024u) { pf0 = pf1 = 0; return; }
            const uint32_t ya = pf0 + 16u * lane, yb = ya + 1024u;
            if (ya < pf1) pfa = *reinterpret_cast<const LZF_GLOBAL u32x4*>(outb + ya);
            if (yb < pf1) pfb = *reinterpret_cast<const LZF_GLOBAL u32x4*>(outb + yb);
        };
        auto prefetch_commit = [&]() {
            if (pf1 > pf0 && pf0 == fp) {
                const uint32_t ya = pf0 + 16u * lane, yb = ya + 1024u;
                if (ya < pf1) *reinterpret_cast<u32x4*>(&ring[ya & kMask]) = pfa;
                if (yb < pf1) *reinterpret_cast<u32x4*>(&ring[yb & kMask]) = pfb;
                fp = pf1;
            }
            pf0 = pf1 = 0;
        };
        // write what the resolver has finished to HBM (whole granules), in ticket order
        auto flush_resolved = [&](uint32_t upto_t) {
            if (flushed_t < upto_t) {                    // (the ends only grow: everything up to the last resolved ticket's in one pass)
                const uint32_t e = __builtin_amdgcn_readlane(endv, (upto_t - 1u) & 63u);
                if (e > fl) { flush_range(fl, e); fl = e; }
                flushed_t = upto_t;
            }
        };
#ifdef LZF_SEG_TIME
        long long tm_swait = 0, tm_fill = 0, tm_old = 0;
        bool gave_up = false;
        auto wait_resolved = [&](uint32_t tt) { const long long t0 = clock64(); if (tt && !gave_up && !flag_wait_above(1, tt - 1u)) gave_up = true; tm_swait += clock64() - t0; };
#else
        bool gave_up = false;
        auto wait_resolved = [&](uint32_t tt) { if (tt && !gave_up && !flag_wait_above(1, tt - 1u)) gave_up = true; };
#endif
        u32x4 rA = u32x4{0, 0, 0, 0}, rB = u32x4{0, 0, 0, 0};
        if (n) { rA = recs[lane < n ? lane : n - 1u]; rB = recs[64u + lane < n ? 64u + lane : n - 1u]; }
        uint32_t prev_end = rb;                          // biased end of the previous sequence
        // Sources older than the ring (class 7) of the NEXT batch's first sub-batch, loaded while this batch is staged: with the small
        // rings every sub-batch of a text block has some (offsets beyond ~20 KiB), and their round trip to HBM sat in the stager's
        // critical path — at four blocks per CU the stager spent 47 % of the block's time there and the resolver a third of its time
        // waiting for tickets (profiles/r05_seg_groups.txt, section 7).  Per lane: a match of 4..32 bytes whose source is below `safe`.
        uint64_t ov0 = 0, ov1 = 0, ov2 = 0, ov3 = 0;
        uint32_t o_for = kNone;                          // the batch (its first record) the registers were loaded for
        bool o_have = false;
        for (uint32_t i0 = 0; i0 < n && !gave_up; i0 += 64u) {
#ifdef LZF_ANALYSIS      // LZF_SEG_FORCE=stager: the stager of every odd job gives up at its third batch
            if (c.dbg_force == 1u && (j & 1u) && i0 >= 128u) { gave_up = true; if (lane == 0u) flag_set(3, 1u); break; }
#endif
            const uint32_t nb = n - i0 < 64u ? n - i0 : 64u;
            const u32x4 r = rA;
            const uint32_t M = lane < nb ? r[0] : 0u, dy = r[1], off = r[3];
            const uint32_t sub = lane < nb ? (r[2] & 0xFFu) : 0xFFu, cls = lane < nb ? r[2] >> 24 : 0u;
            const uint32_t nsub = __builtin_amdgcn_readlane(r[2] & 0xFFu, (nb - 1u) & 63u) + 1u;
            asm volatile("" ::: "memory");
            rA = rB;
            { const uint32_t ix = i0 + 128u + lane; rB = recs[ix < n ? ix : n - 1u]; }
            asm volatile("" ::: "memory");
            const uint32_t endy = dy + M;                // biased end of the sequence
            for (uint32_t s_i = 0; s_i < nsub && !gave_up; ++s_i) {
                const unsigned long long msub = __ballot(sub == s_i);
                if (!msub) { gave_up = true; if (lane == 0u) flag_set(3, 1u); break; }      // (records that do not number their sub-batches 0, 1, 2 ...: not ours)
                const uint32_t last = 63u - (uint32_t)__builtin_clzll(msub);
                const uint32_t oe = __builtin_amdgcn_readlane(endy, last);      // biased end of the sub-batch
                if (const unsigned long long mg8 = __ballot(sub == s_i && cls == 8u)) {
                    // ---- a sequence larger than a sub-batch: everything in front of it resolved and in HBM, then HBM -> HBM
                    // (it is alone in its sub-batch but for the empty sequences that pad a tile's last batch)
                    const uint32_t gl = (uint32_t)__builtin_ctzll(mg8);
                    const uint32_t g_dy = __builtin_amdgcn_readlane(dy, gl), g_M = __builtin_amdgcn_readlane(M, gl), g_off = __builtin_amdgcn_readlane(off, gl);
                    wait_resolved(ticket);
                    flush_resolved(ticket);
                    pf0 = pf1 = 0;                        // (what was loaded early may lie under the copy)
                    if (fp < ((prev_end + 15u) & ~15u)) fill_to((prev_end + 15u) & ~15u);
                    flush_range(fl, prev_end); if (prev_end > fl) fl = prev_end;
                    wave_store_fence();
                    if (g_M) {
                        gu8* const src = outb + (g_dy - g_off);
                        uint32_t pos = 0, av = g_off;
                        while (pos < g_M) {               // the same doubling as ring_match, through HBM
                            const uint32_t cc = av < g_M - pos ? av : g_M - pos;
                            wave_copy_long(outb + g_dy + pos, src, cc, lane);
                            pos += cc; av += cc;
                            if (pos < g_M) wave_store_fence();
                        }
                        wave_store_fence();
                    }
                    // the ring's history is read back from HBM: [oe - (R - fetch-ahead), oe), granules
                    fl = oe; safe = oe;
                    {
                        const uint32_t keep = (uint32_t)R - (NS * kSpan + 128u);
                        const uint32_t g1 = (oe + 15u) & ~15u;
                        fp = g1 > keep ? (g1 - keep) & ~15u : 0u;
                        // (the last granule may hold bytes of the next sequence: they are in `out` already when literals, holes otherwise)
                        fill_to(g1);
                    }
                } else {
                    // ---- sub-batch: the ring filled, the sources older than the ring moved in
                    // Room: fewer than NT tickets unresolved, and the ring filled no further than two spans beyond the end of the oldest
                    // unresolved ticket r (with the 3 KiB the early loads may add: within the fetch-ahead the records stage assumed,
                    // oe_r + 3 spans + 64 — with three tickets in flight that held by itself, a sub-batch being at most a span).
                    uint32_t rs = 0;
                    {
                        const uint32_t need = (oe + 15u) & ~15u;
#ifdef LZF_SEG_TIME
                        const long long t0 = clock64();
#endif
                        for (uint32_t spin = 0; !gave_up; ++spin) {
                            rs = flag_get(1);
                            const uint32_t er = rs < ticket ? __builtin_amdgcn_readlane(endv, rs & 63u) : need;
                            if (ticket - rs < NT && need <= er + 2u * kSpan + 16u) break;
                            if ((spin & 63u) == 63u && (flag_get(3) != 0u || spin > (1u << 22))) { if (lane == 0u) flag_set(3, 1u); gave_up = true; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
#ifdef LZF_SEG_TIME
                        tm_swait += clock64() - t0;
#endif
                    }
#if defined(LZF_SEG_DBG_SKIP) && LZF_SEG_DBG_SKIP == 3      // analysis: a stager that moves no bytes (what the resolver costs on its own)
                    if (false)
#endif
                    flush_resolved(rs < ticket ? rs : ticket);
#if defined(LZF_SEG_DBG_SKIP) && LZF_SEG_DBG_SKIP == 3
                    fp = (oe + 15u) & ~15u;
#endif
#ifdef LZF_SEG_TIME
                    const long long tf0 = clock64();
#endif
                    prefetch_commit();
                    fill_to((oe + 15u) & ~15u);
#ifdef LZF_SEG_TIME
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    tm_fill += clock64() - tf0;
#endif
                    prefetch_issue((oe + 15u) & ~15u);
                    bool old = sub == s_i && cls == 7u;
#ifdef LZF_SEG_TIME
                    const long long to0 = clock64();
#endif
                    {   // the lanes whose bytes came early (two-ended pieces, as put_small_glb stores them)
                        const bool early = old && s_i == 0u && o_for == i0 && o_have;
                        if (__ballot(early)) {
                            if (early) {
                                const uint32_t d = ring_a + (dy & kMask);
                                if (M >= 8u) {
                                    lds_st64(d, ov0);
                                    if (M > 16u) { lds_st64(d + 8u, ov1); lds_st64(d + M - 16u, ov2); }
                                    lds_st64(d + M - 8u, ov3);
                                } else { lds_st32(d, (uint32_t)ov0); lds_st32(d + M - 4u, (uint32_t)ov3); }
                            }
                            old = old && !early;
                        }
                    }
                    if (__ballot(old)) {
                        // (a rare path: its operands pass through an empty asm statement, or its predicates are computed per batch)
                        uint32_t M_o = M, off_o = off, dy_o = dy;
                        asm volatile("" : "+v"(M_o), "+v"(off_o), "+v"(dy_o));
                        const uint32_t M = M_o, off = off_o, dy = dy_o;
                        const uint32_t span = M < off ? M : off;
                        const uint32_t sy = dy - off;
                        if (__ballot(old && sy + span > fl)) { wait_resolved(ticket); flush_resolved(ticket); }
                        if (__ballot(old && (sy + span > fl ? fl : sy + span) > safe)) { wave_store_fence(); safe = fl; }
                        const uint32_t di = dy & kMask;
                        const bool own = old && sy + span <= fl && M <= 64u && off >= M && di + M <= (uint32_t)R;
                        if (own) put_small_glb(ring_a + di, outb + sy, M);
                        for (unsigned long long m = __ballot(old && !own); m; m &= m - 1ull) {
                            const uint32_t q = (uint32_t)__builtin_ctzll(m);
                            const uint32_t qM = __builtin_amdgcn_readlane(M, q), qs = __builtin_amdgcn_readlane(sy, q), qd = __builtin_amdgcn_readlane(dy, q);
                            const uint32_t qo = __builtin_amdgcn_readlane(off, q);
                            // (all lanes, a byte each per step: the source lies in front of the destination, so no byte read here is one written here)
                            for (uint32_t i = lane; i < qM; i += kWave) {
                                const uint32_t y = qs + (qo < qM ? i % qo : i);
                                ring[(qd + i) & kMask] = y < fl ? (uint8_t)outb[y] : ring[y & kMask];
                            }
                        }
                    }
#ifdef LZF_SEG_TIME
                    { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tm_old += clock64() - to0; }
#endif
                }
                const uint32_t eg = oe & ~15u;
                if (lane == (ticket & 63u)) endv = eg;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the ring and the slot written; loads and stores to HBM stay in flight
                ++ticket;
                if (lane == 0u) flag_set(0, ticket);
                prev_end = oe;
            }
            // ---- the next batch's old sources, early (its records are in rA)
            o_for = kNone; o_have = false;
            if (i0 + 64u < n && !gave_up) {
                const u32x4 q = rA;
                const uint32_t nbn = n - (i0 + 64u) < 64u ? n - (i0 + 64u) : 64u;
                const uint32_t Mn = q[0], dyn = q[1], offn = q[3];
                const uint32_t syn = dyn - offn, din = dyn & kMask;
                o_have = lane < nbn && (q[2] >> 24) == 7u && (q[2] & 0xFFu) == 0u && Mn >= 4u && Mn <= 32u && offn >= Mn &&
                         din + Mn <= (uint32_t)R && syn + Mn <= safe && syn + Mn <= fl;
                o_for = i0 + 64u;
                if (o_have) {
                    cgu8* g = outb + syn;
                    if (Mn >= 8u) { ov0 = ld8(g); ov3 = ld8(g + Mn - 8u); if (Mn > 16u) { ov1 = ld8(g + 8u); ov2 = ld8(g + Mn - 16u); } }
                    else { ov0 = ld4(g); ov3 = ld4(g + Mn - 4u); }
                }
            }
        }
        wait_resolved(ticket);
        // A pair that gave up — this wave's bounded wait, the resolver's (ctl[3]), records this stage does not handle — leaves the
        // job alone: no result, done stays 0, and the pair kernel launched behind this one decodes the block from its first byte.
        if (gave_up || flag_get(3) != 0u) return;
#if !(defined(LZF_SEG_DBG_SKIP) && LZF_SEG_DBG_SKIP == 3)
        flush_resolved(ticket);
        flush_range(fl, total + rb);
#endif
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

#ifdef LZF_SEG_TIME
        if (lane == 0u) reinterpret_cast<uint16_t*>(&c.st[j].pad)[1] = (uint16_t)(tm_swait >> 14);
#endif
        if (lane == 0u) {
            c.results[j].out_len = total;
            c.results[j].status = LZF_OK;
#ifdef LZF_SEG_TIME
            // resolver's wait for tickets | stager: fills | stager: sources older than the ring | total — a byte each, in 2^17 cycles
            { auto b8 = [](long long v) -> uint32_t { const long long x = v >> 17; return x > 255 ? 255u : (uint32_t)x; };
              c.results[j].reserved = b8((long long)flag_get(2) << 10) | (b8(c.dbg_force == 8u ? tm_swait : tm_fill) << 8) | (b8(tm_old) << 16) | (b8(clock64() - t_start) << 24); }
#else
            c.results[j].reserved = (uint32_t)((clock64() - t_start) >> 10);
#endif
            c.st[j].done = 1u;
        }
    }
}
template __global__ void lzf_seg_resolve_pair_kernel<32768>(seg_ctx);
template __global__ void lzf_seg_resolve_pair_kernel<65536>(seg_ctx);
template __global__ void lzf_seg_resolve_pair_kernel<131072>(seg_ctx);

}  // namespace lzf
