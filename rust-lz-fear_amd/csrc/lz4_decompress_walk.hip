// lz4_decompress_walk.hip — fourth-generation raw::decompress_raw kernel (src/raw/decompress.rs:58-138):
// a producer / consumer pair of wavefronts per block like lz4_decompress_paired.hip, with a parser that
// walks the token chain instead of tabulating it.
//
// PARSE.  A chunk is 64 regions of S compressed bytes (S = 64..128), staged in LDS once.  Every lane walks the
// token chain of its own region serially, ONE unaligned 8-byte LDS read per token (the read starts one byte
// before the token, so it also delivers the match-length extension byte of the previous token: that is the only
// byte of a plain token that lies behind a data-dependent position).  Region starts are not token starts, so:
//   pass 0   every lane walks from its region start (lane 0 from the chunk's true first token), marks the token
//            positions it visits in a per-lane bit mask (registers) and keeps its exit (first position >= region end);
//   pass k   start[i] = max(exit[0..i-1]) (DPP prefix max).  A lane whose start changed walks from it until it
//            steps on a marked position — from there on the chain is the pass-0 chain, whose exit and token count
//            (population count of the mask) are known — or leaves the region.  Until no start changes; lane 0 is
//            true from the beginning, so the fixed point is the true chain.  Large regions break the dependence of
//            an exit on its entry quickly: 2–4 passes on text, ~10 on streams of 3-byte sequences (tools/seq_stats.c).
//   record   ranks by DPP prefix sum of the token counts, then one more walk from the true starts writes the
//            32-bit token entries (chunk offset | L << 16 | (M - 4) << 24, as the copy stage expects) in stream order.
// Tokens a plain hop cannot express (a 0xFF length byte, a body that leaves the staged bytes, the last 24 bytes of the
// input) park their lane; parked lanes are served together by the general routine (decompress.rs:61-71 without the
// copies) and get an "escape" entry that the copy stage decodes in full.
//
// COPY.  lz4_decompress_batch_phase.inc (shared with the other batched kernels).
#include "lzf_device.h"
#include "kernels.h"
#include "lzf_copy_helpers.h"
#include <type_traits>

namespace lzf {
namespace {
__device__ __forceinline__ uint64_t lds_ld64(uint32_t a) {
    uint64_t v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); return v;
}
}  // namespace

template <int RING, int S, int TOKCAP>
__global__ __launch_bounds__(128) void lzf_decompress_walk_kernel(
    const lzf_decompress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs,
    const uint32_t* __restrict__ perm) {
    constexpr bool STAGE = true;
    constexpr uint32_t kMask = RING - 1;
    constexpr uint32_t kSpanMax = RING / 3;            // output bytes one batch may produce
    constexpr uint32_t kNearHist = RING - kSpanMax;    // history before the batch that stays intact in the ring
    constexpr uint32_t kChunk = 64u * S;               // compressed bytes whose tokens one parse covers
    constexpr uint32_t kCB = kChunk + 64u;             // staged bytes: the chunk + room for token bodies
    constexpr uint32_t kFront = 16u;                   // bytes in front of the staged chunk (a hop reads from position - 1)
    constexpr uint32_t kBuf = kFront + kCB;
    constexpr uint32_t NM = ((uint32_t)S + 63u) / 64u; // 64-bit words of a lane's mark mask
    static_assert(kChunk <= 65536, "token positions are stored as u16 offsets into the chunk");
    static_assert(kCB % 16 == 0 && S % 4 == 0 && S <= 128, "chunk buffer is filled in 16-byte pieces");
    constexpr uint32_t kTokWords = (uint32_t)TOKCAP + 64u;
    __shared__ __attribute__((aligned(16))) uint8_t ring[RING];
    __shared__ __attribute__((aligned(16))) uint8_t cbufs[2u * kBuf];      // double-buffered staged chunk
    __shared__ __attribute__((aligned(16))) uint32_t tokl[2u * kTokWords]; // double-buffered token list
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
        const uint32_t rb = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u);   // ring bias
        const uint32_t ring_a = lds_addr(ring);
#define RIDX(x) (((x) + rb) & kMask)

        // ring <- out[a, b)   (b - a <= RING; caller made out[a,b) visible)
        auto ring_fill = [&](uint32_t a, uint32_t b) {
            uint32_t nh = (16u - ((a + rb) & 15u)) & 15u; if (nh > b - a) nh = b - a;
            if (lane < nh) ring[RIDX(a + lane)] = out[a + lane];
            a += nh;
            const uint32_t nchunks = (b - a) >> 4;
            for (uint32_t c = lane; c < nchunks; c += kWave)
                *reinterpret_cast<u32x4*>(&ring[RIDX(a + 16u * c)]) = *reinterpret_cast<const LZF_GLOBAL u32x4*>(out + a + 16u * c);
            a += nchunks << 4;
            if (lane < b - a) ring[RIDX(a + lane)] = out[a + lane];
        };
        // out[a, b) <- ring
        auto ring_flush = [&](uint32_t a, uint32_t b) {
            uint32_t nh = (16u - ((a + rb) & 15u)) & 15u; if (nh > b - a) nh = b - a;
            if (lane < nh) out[a + lane] = ring[RIDX(a + lane)];
            a += nh;
            const uint32_t nchunks = (b - a) >> 4;
            for (uint32_t c = lane; c < nchunks; c += kWave)
                *reinterpret_cast<LZF_GLOBAL u32x4*>(out + a + 16u * c) = *reinterpret_cast<const u32x4*>(&ring[RIDX(a + 16u * c)]);
            a += nchunks << 4;
            if (lane < b - a) out[a + lane] = ring[RIDX(a + lane)];
        };

#define PHASE(i) do { } while (0)
        if (threadIdx.x == 0) ctl_stop = 0;
        __syncthreads();
        if (role == 0u) {
            // ================================ PARSER ================================
            uint32_t cstart = 0;                 // a true token position (or len)
            for (uint32_t kc = 0;; ++kc) {
                const uint32_t bsel = kc & 1u;
                uint8_t* const cbuf = cbufs + bsel * kBuf + kFront;
                uint32_t* const toks = tokl + bsel * kTokWords;
                const uint32_t cbuf_a = lds_addr(cbufs) + bsel * kBuf + kFront, toks_a = lds_addr(tokl) + bsel * kTokWords * 4u;
                const bool valid = cstart < len && *(volatile int*)&ctl_stop == 0;
                uint32_t cend_next = cstart;
                if (valid) {
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
                if (lane == 0u) ctl_valid[bsel] = valid ? 1 : 0;
                __syncthreads();                 // chunk kc is parsed (and the copier is done with chunk kc - 1)
                if (!valid) break;
                cstart = cend_next;
            }
        } else {
            // ================================ COPIER ================================
            o = (uint32_t)job.out_existing_len;
            uint32_t safe = o;   // out[0, safe) is visible to this wave's global loads
            if (o > 0) ring_fill(o > (uint32_t)RING ? o - RING : 0u, o);   // Vec content on entry = history
            for (uint32_t kc = 0;; ++kc) {
                __syncthreads();                 // chunk kc is parsed
                const uint32_t bsel = kc & 1u;
                if (*(volatile int*)&ctl_valid[bsel] == 0) break;
                if (status != LZF_OK) continue;  // keep meeting the parser until it sees the stop flag
                const uint32_t* const toks = tokl + bsel * kTokWords;
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

#define LZF_TOKEN_AT(i) (toks[(i)] & 0xFFFFu)
#define LZF_TOKEN_WORD(i) toks[(i)]
#include "lz4_decompress_batch_phase.inc"
#undef LZF_TOKEN_WORD
#undef LZF_TOKEN_AT
                if (status == LZF_OK && cerr != LZF_OK) status = cerr;
                if (status != LZF_OK && lane == 0u) *(volatile int*)&ctl_stop = 1;
            }
        }
#undef PHASE
#undef RIDX
    }
    if (role == 1u && lane == 0u) {
        results[jid].out_len = o;
        results[jid].status = status;
        results[jid].reserved = (uint32_t)((clock64() - t_start) >> 10);   // diagnostic: shader kilo-cycles spent on this job
    }
}

#define LZF_INSTK(NAME, RG, S_, T) template __global__ void lzf_decompress_walk_kernel<RG, S_, T>(const lzf_decompress_job*, lzf_job_result*, uint32_t, const uint32_t*);
LZF_WALK_VARIANTS(LZF_INSTK)
#undef LZF_INSTK

}  // namespace lzf
