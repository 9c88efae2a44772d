// lz4_decompress_v4.hip — raw::decompress_raw (src/raw/decompress.rs:58-138) for gfx950: a producer / consumer pair of
// wavefronts per block.  Wave 0 PARSES chunk k+1 of the compressed input while wave 1 COPIES chunk k; the staged chunk
// and the token list are double-buffered in LDS and the waves meet at one barrier per chunk.
//   PARSER = 0  tabulating parse (lz4_decompress_parse_phase.inc: nxt[] / ex[] tables over 16..48-byte regions)
//   PARSER = 1  region-walk parse (lz4_decompress_walk_phase.inc: serial token walks over 48..128-byte regions)
//   copy stage  lz4_decompress_copy2.inc: linear LDS window, every match of a batch copied at once and repeated until stable
// Error precedence is the reference's: within a sequence literal EOF / LSIC EOF (UnexpectedEnd), MemoryLimitExceeded,
// ZeroDeduplicationOffset, InvalidDeduplicationOffset (decompress.rs:63-75,82-89); across sequences the first in stream order.
#include "lzf_device.h"
#include "kernels.h"
#include "lzf_copy_helpers.h"
#include "lzf_parse_helpers.h"
#include <type_traits>

namespace lzf {
namespace {
__device__ __forceinline__ uint64_t lds_ld64(uint32_t a) {
    uint64_t v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); return v;
}
}  // namespace

template <int W, int S, int TOKCAP, int PARSER>
__global__ __launch_bounds__(128) void lzf_decompress_v4_kernel(
    const lzf_decompress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs,
    const uint32_t* __restrict__ perm) {
    constexpr bool STAGE = true;
    constexpr int SPAN = W / 4;                        // output bytes one batch may produce
    constexpr int HKEEP = W / 2;                       // history a slide keeps
    static_assert(HKEEP + SPAN + 96 <= W && W % 1024 == 0, "a batch fits behind the kept history");
    constexpr uint32_t kChunk = 64u * S;               // compressed bytes whose tokens one parse covers
    constexpr uint32_t kCB = kChunk + 64u;             // staged bytes: the chunk + room for token bodies
    constexpr uint32_t kFront = 16u;                   // addressable bytes in front of (and slack behind) a staged chunk
    constexpr uint32_t kBuf = kFront + kCB;
    constexpr uint32_t NM = ((uint32_t)S + 63u) / 64u; // 64-bit words of a lane's mark mask (walk parser)
    static_assert(kChunk <= 65536, "token positions are stored as u16 offsets into the chunk");
    static_assert(kCB % 16 == 0 && S % 4 == 0 && S <= 128, "chunk buffer is filled in 16-byte pieces, tables in dwords");
    constexpr uint32_t kExStride = (uint32_t)S + 4u;
    constexpr uint32_t kTokBytes = ((uint32_t)TOKCAP + 64u) * 4u;     // 32-bit entries: offset | L << 16 | (M - 4) << 24
    constexpr uint32_t kExBytes = PARSER == 0 ? 64u * kExStride : 0u;
    constexpr uint32_t kTokex = ((kTokBytes > kExBytes ? kTokBytes : kExBytes) + 15u) & ~15u;
    __shared__ __attribute__((aligned(16))) uint8_t win[W + 32 + 512 + 16];        // window + per-lane scratch words
    __shared__ __attribute__((aligned(16))) uint8_t cbufs[2u * kBuf + kFront];      // double-buffered staged chunk
    __shared__ __attribute__((aligned(16))) uint8_t nxt[PARSER == 0 ? kChunk : 16u]; // tabulating parser only
    __shared__ __attribute__((aligned(16))) uint8_t tokexs[2u * kTokex];            // (ex[] while parsing, then) the token list
    __shared__ uint32_t ctl_T[2], ctl_cstart[2];
    __shared__ int ctl_err[2], ctl_valid[2], ctl_stop;

    if (blockIdx.x >= n_jobs) return;
    const uint32_t jid = perm ? perm[blockIdx.x] : blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0: parser, 1: copier (uniform per wavefront)
    const lzf_decompress_job job = jobs[jid];
    const long long t_start = clock64();

    int status = LZF_OK;
    uint32_t o = 0;
    if (job.input_len >= kMaxPosB || job.out_existing_len >= kMaxPosB || job.prefix_len >= kMaxPosB || job.out_existing_len > job.out_cap) {
        status = LZF_CONTRACT;                         // (uniform over the workgroup: no barrier is reached)
    } else {
        cgu8* __restrict__ in = as_global(job.input);
        cgu8* __restrict__ prefix = as_global(job.prefix);
        gu8* out = as_global(job.out);
        const uint32_t len = (uint32_t)job.input_len;
        const uint32_t plen = (uint32_t)job.prefix_len;
        const uint32_t cap = job.out_cap > kMaxPosB ? kMaxPosB : (uint32_t)job.out_cap;
        const uint64_t limit = job.output_limit;
        const uint32_t rb = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u);   // out + x is 16-byte aligned when (x + rb) % 16 == 0

#define PHASE(i) do { } while (0)
        if (threadIdx.x == 0) ctl_stop = 0;
        __syncthreads();
        if (role == 0u) {
            // ================================ PARSER ================================
            const uint32_t nxt_a = lds_addr(nxt);
            uint32_t cstart = 0;                 // a true token position (or len)
            for (uint32_t kc = 0;; ++kc) {
                const uint32_t bsel = kc & 1u;
                uint8_t* const cbuf = cbufs + bsel * kBuf + kFront;
                uint8_t* const tokex = tokexs + bsel * kTokex;
                uint32_t* const toks = reinterpret_cast<uint32_t*>(tokex);
                const uint32_t cbuf_a = lds_addr(cbufs) + bsel * kBuf + kFront, ex_a = lds_addr(tokexs) + bsel * kTokex;
                const uint32_t toks_a = ex_a;
                const bool valid = cstart < len && *(volatile int*)&ctl_stop == 0;
                uint32_t cend_next = cstart;
                if (valid) {
                    if constexpr (PARSER == 0) {
#define LZF_TOK_T uint32_t
#define LZF_THOP_RECORD thop_loop_record32
#include "lz4_decompress_parse_phase.inc"
#undef LZF_THOP_RECORD
#undef LZF_TOK_T
                        // Lengths of the listed tokens, 64 at a time, so that the copier does not have to re-read them:
                        // entry = chunk offset | L << 16 | (M - 4) << 24; L = 255 / M - 4 = 255: not expressible (the copier
                        // decodes that token itself); M - 4 = 254: the block's last sequence (no match).
                        for (uint32_t t0 = 0; t0 < Tc; t0 += kWave) {
                            const uint32_t t = t0 + lane;
                            if (t < Tc) {
                                const uint32_t pos = toks[t] & 0xFFFFu;
                                const uint32_t tp = cstart + pos;
                                const uint32_t w = rd4(tp);
                                uint32_t L = (w >> 4) & 15u, q = tp + 1u, Lc, Mc = 255u;
                                if (L == 15u) { L += (w >> 8) & 255u; ++q; }
                                Lc = L < 255u && !(((w >> 4) & 15u) == 15u && ((w >> 8) & 255u) == 255u) ? L : 255u;
                                if (Lc != 255u) {
                                    q += L;
                                    if (len - q < 2u) Mc = 254u;                 // :70 read_u16 fails: last literals
                                    else {
                                        uint32_t M = w & 15u;
                                        if (M == 15u) { const uint32_t m1 = rdb(q + 2u); M = m1 < 239u ? 15u + m1 : 255u; }
                                        Mc = M;
                                    }
                                }
                                toks[t] = pos | (Lc << 16) | (Mc << 24);
                            }
                        }
                        if (lane == 0u) { ctl_T[bsel] = Tc; ctl_cstart[bsel] = cstart; ctl_err[bsel] = cerr; }
                        cend_next = cend;
                    } else {
                        // ---- stage in[cstart, cstart + kCB) in LDS (zeros beyond the input)
                        {
                            const uint32_t avail = len - cstart < kCB ? len - cstart : kCB;
                            cgu8* g = in + cstart;
#pragma unroll 1
                            for (uint32_t base = 0; base < kCB; base += 4u * 1024u) {
                                u32x4 v[4];
#pragma unroll
                                for (uint32_t k = 0; k < 4u; ++k) {
                                    const uint32_t i = base + k * 1024u + lane * 16u;
                                    v[k] = u32x4{0, 0, 0, 0};
                                    if (i + 16u <= avail) v[k] = ld16(g + i);
                                    else if (i < avail) { for (uint32_t t = 0; i + t < avail; ++t) v[k][(t >> 2) & 3u] |= (uint32_t)g[i + t] << ((t & 3u) * 8u); }
                                }
#pragma unroll
                                for (uint32_t k = 0; k < 4u; ++k) {
                                    const uint32_t i = base + k * 1024u + lane * 16u;
                                    if (i < kCB) *reinterpret_cast<u32x4*>(&cbuf[i]) = v[k];
                                }
                            }
                        }
#include "lz4_decompress_walk_phase.inc"
                        if (lane == 0u) { ctl_T[bsel] = Tc; ctl_cstart[bsel] = cstart; ctl_err[bsel] = cerr; }
                        cend_next = cend;
                    }
                }
                if (lane == 0u) ctl_valid[bsel] = valid ? 1 : 0;
                __syncthreads();                 // chunk kc is parsed (and the copier is done with chunk kc - 1)
                if (!valid) break;
                cstart = cend_next;
            }
        } else {
            // ================================ COPIER ================================
            const uint32_t win_a = lds_addr(win);
            auto AL = [&](uint32_t x) -> uint32_t { return ((x + rb) & ~15u) - rb; };      // granule boundary at or below x
            const uint32_t lim32 = limit < (uint64_t)cap ? (uint32_t)limit : cap;     // a match may end at lim32 at most
            uint32_t wlo = 0, hlo = 0, fl = 0;   // window origin; lowest position the window holds; out[0, fl) is in HBM
            // window <- out[a, b)   (caller made out[a, b) visible; b - wlo <= W)
            auto win_fill = [&](uint32_t a, uint32_t b) {
                uint32_t nh = (16u - ((a + rb) & 15u)) & 15u; if (nh > b - a) nh = b - a;
                if (lane < nh) win[a - wlo + lane] = out[a + lane];
                a += nh;
                const uint32_t nchunks = (b - a) >> 4;
                for (uint32_t c = lane; c < nchunks; c += kWave)
                    *reinterpret_cast<u32x4*>(&win[a - wlo + 16u * c]) = *reinterpret_cast<const LZF_GLOBAL u32x4*>(out + a + 16u * c);
                a += nchunks << 4;
                if (lane < b - a) win[a - wlo + lane] = out[a + lane];
            };
            // out[a, b) <- window
            auto win_flush = [&](uint32_t a, uint32_t b) {
                uint32_t nh = (16u - ((a + rb) & 15u)) & 15u; if (nh > b - a) nh = b - a;
                if (nh) { if (lane < nh) out[a + lane] = win[a - wlo + lane]; a += nh; }
                const uint32_t nchunks = (b - a) >> 4;
                for (uint32_t c = lane; c < nchunks; c += kWave)
                    *reinterpret_cast<LZF_GLOBAL u32x4*>(out + a + 16u * c) = *reinterpret_cast<const u32x4*>(&win[a - wlo + 16u * c]);
                a += nchunks << 4;
                if (lane < b - a) out[a + lane] = win[a - wlo + lane];
            };
            o = (uint32_t)job.out_existing_len;
            uint32_t safe = o;   // out[0, safe) is visible to this wave's global loads
            hlo = o > (uint32_t)HKEEP ? o - (uint32_t)HKEEP : 0u;
            wlo = AL(hlo);
            if (o > hlo) win_fill(hlo, o);       // Vec content on entry = history
            fl = o;
            for (uint32_t kc = 0;; ++kc) {
                __syncthreads();                 // chunk kc is parsed
                const uint32_t bsel = kc & 1u;
                if (*(volatile int*)&ctl_valid[bsel] == 0) break;
                if (status != LZF_OK) continue;  // keep meeting the parser until it sees the stop flag
                const uint32_t* const toks = reinterpret_cast<const uint32_t*>(tokexs + bsel * kTokex);
                const uint32_t cbuf_a = lds_addr(cbufs) + bsel * kBuf + kFront;
                const uint32_t cstart = *(volatile uint32_t*)&ctl_cstart[bsel];
                const uint32_t Tc = *(volatile uint32_t*)&ctl_T[bsel];
                const int cerr = *(volatile int*)&ctl_err[bsel];
                auto rdb = [&](uint32_t q) -> uint32_t {
                    const uint32_t r = q - cstart;
                    if (r < kCB) return lds_ld8(cbuf_a + r);
                    return (uint32_t)in[q];
                };
                // 4 input bytes at q (missing bytes past the end read as 0)
                auto rd4 = [&](uint32_t q) -> uint32_t {
                    const uint32_t r = q - cstart;
                    if (r + 4u <= kCB) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cbuf_a + r) : "memory"); return v; }
                    uint32_t v = 0;
                    for (uint32_t i = 0; i < 4u && q + i < len; ++i) v |= rdb(q + i) << (8u * i);
                    return v;
                };
#include "lz4_decompress_copy2.inc"
                if (status == LZF_OK && cerr != LZF_OK) status = cerr;
                if (status != LZF_OK && lane == 0u) *(volatile int*)&ctl_stop = 1;
            }
            win_flush(fl, o);                    // the last partial granule
        }
#undef PHASE
    }
    if (role == 1u && lane == 0u) {
        results[jid].out_len = o;
        results[jid].status = status;
        results[jid].reserved = (uint32_t)((clock64() - t_start) >> 10);   // diagnostic: shader kilo-cycles spent on this job
    }
}

#define LZF_INST4(NAME, W_, S_, T, P) template __global__ void lzf_decompress_v4_kernel<W_, S_, T, P>(const lzf_decompress_job*, lzf_job_result*, uint32_t, const uint32_t*);
LZF_V4_VARIANTS(LZF_INST4)
#undef LZF_INST4

}  // namespace lzf
