// lz4_decompress_paired.hip — the staged decompress kernel (lz4_decompress_batched.hip, STAGE form) as a
// producer / consumer pair: a workgroup of two wavefronts per block, wave 0 PARSES chunk k+1 while wave 1
// COPIES chunk k.  Same contract (src/raw/decompress.rs:58-138), same parse (staged chunk, nxt[] and ex[] tables,
// fixed point over the region starts, hand-scheduled record pass) and the same copy stage
// (lz4_decompress_batch_phase.inc); the chunk buffer and the token list are double-buffered in LDS and the two
// waves meet at one barrier per chunk.  A block's latency becomes max(parse, copy) instead of their sum, and a
// wave's share of LDS drops to ~5 KB, so a CU holds 28 waves instead of 21.
#include "lzf_device.h"
#include "kernels.h"
#include "lzf_copy_helpers.h"
#include "lzf_parse_helpers.h"

namespace lzf {

template <int RING, int S, int TOKCAP>
__global__ __launch_bounds__(128) void lzf_decompress_paired_kernel(
    const lzf_decompress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs,
    const uint32_t* __restrict__ perm, const seg_job* __restrict__ done) {
    constexpr bool STAGE = true;
    constexpr uint32_t kMask = RING - 1;
    constexpr uint32_t kSpanMax = RING / 3;            // output bytes one batch may produce
    constexpr uint32_t kNearHist = RING - kSpanMax;    // history before the batch that stays intact in the ring
    constexpr uint32_t kChunk = 64u * S;               // compressed bytes whose tokens one parse covers
    constexpr uint32_t kCB = kChunk + 64u;             // staged bytes: the chunk + room for token bodies
    static_assert(kChunk <= 65536, "token positions are stored as u16 offsets into the chunk");
    static_assert(kCB % 16 == 0 && S % 4 == 0, "chunk buffer is filled in 16-byte pieces, tables in dwords");
    constexpr uint32_t kExStride = (uint32_t)S + 4u;
    constexpr uint32_t kTokBytes = ((uint32_t)TOKCAP + 64u) * 4u;     // 32-bit entries: offset | L << 16 | (M - 4) << 24
    constexpr uint32_t kExBytes = 64u * kExStride;
    constexpr uint32_t kTokex = ((kTokBytes > kExBytes ? kTokBytes : kExBytes) + 15u) & ~15u;
    __shared__ __attribute__((aligned(16))) uint8_t ring[RING];
    __shared__ __attribute__((aligned(16))) uint8_t cbufs[2u * kCB];       // double-buffered staged chunk
    __shared__ __attribute__((aligned(16))) uint8_t nxt[kChunk];           // parser only
    __shared__ __attribute__((aligned(16))) uint8_t tokexs[2u * kTokex];   // ex[] while parsing, then the token list
    __shared__ uint32_t ctl_T[2], ctl_cstart[2];
    __shared__ int ctl_err[2], ctl_valid[2], ctl_stop;

    if (blockIdx.x >= n_jobs) return;
    const uint32_t jid = perm ? perm[blockIdx.x] : blockIdx.x;
    if (done && done[jid].done) return;               // finished by the segmented pipeline (uniform over the workgroup)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0: parser, 1: copier (uniform per wavefront)
    const lzf_decompress_job job = jobs[jid];
    const long long t_start = clock64();

    int status = LZF_OK;
    uint32_t o = 0;
#ifdef LZF_DBG_PHASE_SEL
    long long ph_acc_out = 0;
#endif
#ifdef LZF_DBG_ROUNDS
    uint32_t dbg_rounds = 0, dbg_batches = 0, dbg_lane_stat = 0;
#endif
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
        const uint32_t ring_a = lds_addr(ring), nxt_a = lds_addr(nxt);
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


#ifdef LZF_DBG_PHASE_SEL   // analysis: cycles the copier spends in section LZF_DBG_PHASE_SEL of its batch loop (section i ends at PHASE(i);
                           // 0 = between batches: loop control, chunk hand-over, waiting for the parser) -> results[].reserved
        long long ph_t = clock64(), ph_acc = 0;
#define PHASE(i) do { const long long tn__ = clock64(); if ((i) == LZF_DBG_PHASE_SEL) ph_acc += tn__ - ph_t; ph_t = tn__; } while (0)
#else
#define PHASE(i) do { } while (0)
#endif
        if (threadIdx.x == 0) ctl_stop = 0;
        __syncthreads();
        if (role == 0u) {
            // ================================ PARSER ================================
            uint32_t cstart = 0;                 // a true token position (or len)
            for (uint32_t kc = 0;; ++kc) {
                const uint32_t bsel = kc & 1u;
                uint8_t* const cbuf = cbufs + bsel * kCB;
                uint8_t* const tokex = tokexs + bsel * kTokex;
                uint32_t* const toks = reinterpret_cast<uint32_t*>(tokex);
                const uint32_t cbuf_a = lds_addr(cbufs) + bsel * kCB, ex_a = lds_addr(tokexs) + bsel * kTokex;
                const bool valid = cstart < len && *(volatile int*)&ctl_stop == 0;
                uint32_t cend_next = cstart;
                if (valid) {
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
                const uint32_t* const toks = reinterpret_cast<const uint32_t*>(tokexs + bsel * kTokex);
                const uint32_t cbuf_a = lds_addr(cbufs) + bsel * kCB;
                const uint32_t cstart = *(volatile uint32_t*)&ctl_cstart[bsel];
                const uint32_t Tc = *(volatile uint32_t*)&ctl_T[bsel];
                const int cerr = *(volatile int*)&ctl_err[bsel];
                // byte of the input at absolute position q >= cstart
                // (asm LDS read on purpose: with two plain loads hipcc selects between the pointers and emits
                //  one FLAT load, which waits on both memory counters at every use)
                auto rdb = [&](uint32_t q) -> uint32_t {
                    const uint32_t r = q - cstart;
                    if (r < kCB) return lds_ld8(cbuf_a + r);
                    return (uint32_t)in[q];
                };
                // One token at p (p < len): position of the next token; false on UnexpectedEnd.
                // decompress.rs:61-71 without the copies.
                // 4 input bytes at q (missing bytes past the end read as 0)
                auto rd4 = [&](uint32_t q) -> uint32_t {
                    const uint32_t r = q - cstart;
                    if (!STAGE) { if (q + 4u <= len) return ld4(in + q); }
                    else if (r + 4u <= kCB) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cbuf_a + r) : "memory"); return v; }
                    uint32_t v = 0;
                    for (uint32_t i = 0; i < 4u && q + i < len; ++i) v |= rdb(q + i) << (8u * i);
                    return v;
                };

#define LZF_TOKEN_AT(i) (toks[(i)] & 0xFFFFu)
#define LZF_TOKEN_WORD(i) toks[(i)]
#define LZF_DBG_ROUNDS_HERE
#include "lz4_decompress_batch_phase.inc"
#undef LZF_DBG_ROUNDS_HERE
#undef LZF_TOKEN_WORD
#undef LZF_TOKEN_AT
                if (status == LZF_OK && cerr != LZF_OK) status = cerr;
                if (status != LZF_OK && lane == 0u) *(volatile int*)&ctl_stop = 1;
            }
        }
#ifdef LZF_DBG_PHASE_SEL
        ph_acc_out = ph_acc;
#endif
#undef PHASE
#undef RIDX
    }
    if (role == 1u && lane == 0u) {
        results[jid].out_len = o;
        results[jid].status = status;
#ifdef LZF_DBG_PHASE_SEL
        results[jid].reserved = (uint32_t)(ph_acc_out >> 10);
#elif defined(LZF_DBG_ROUNDS)
        results[jid].reserved = LZF_DBG_ROUNDS == 1 ? dbg_rounds : LZF_DBG_ROUNDS == 2 ? dbg_batches : dbg_lane_stat;
#else
        results[jid].reserved = (uint32_t)((clock64() - t_start) >> 10);   // diagnostic: shader kilo-cycles spent on this job
#endif
    }
}

#define LZF_INSTP(NAME, RG, S_, T) template __global__ void lzf_decompress_paired_kernel<RG, S_, T>(const lzf_decompress_job*, lzf_job_result*, uint32_t, const uint32_t*, const seg_job*);
LZF_PAIRED_VARIANTS(LZF_INSTP)
#undef LZF_INSTP

}  // namespace lzf
