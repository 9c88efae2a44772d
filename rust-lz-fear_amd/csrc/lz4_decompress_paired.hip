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
    const lzf_decompress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs) {
    constexpr bool STAGE = true;
    constexpr uint32_t kMask = RING - 1;
    constexpr uint32_t kSpanMax = RING / 4;            // output bytes one batch may produce
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

    const uint32_t jid = blockIdx.x;
    if (jid >= n_jobs) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0: parser, 1: copier (uniform per wavefront)
    const lzf_decompress_job job = jobs[jid];
    const long long t_start = clock64();

    int status = LZF_OK;
    uint32_t o = 0;
    if (job.input_len >= kMaxPosB || job.out_existing_len >= kMaxPosB || job.prefix_len >= kMaxPosB) {
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


#define PHASE(i) do { } while (0)
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
                uint32_t cend = cstart;
                if (valid) {
                // =====================================================================
                // A0. stage in[cstart, cstart + kCB) in LDS
                // =====================================================================
                if (STAGE) {
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
                // =====================================================================
                // A1. next-token table of the chunk (decompress.rs:61-71 without the copies, plain tokens only)
                // =====================================================================
                if (STAGE) {
                    const uint32_t fast_end = len > 24u ? len - 24u : 0u;
                    const uint32_t fe = fast_end < cstart + kCB ? fast_end : cstart + kCB;     // a plain token's body ends below fe
    #pragma unroll 1
                    for (uint32_t j = lane * 4u; j < kChunk; j += 4u * kWave) {
                        const uint32_t w0 = *reinterpret_cast<const uint32_t*>(&cbuf[j]);
                        const uint32_t w1 = *reinterpret_cast<const uint32_t*>(&cbuf[j + 4u]);
                        uint32_t d[4], qa[4], m1[4]; bool bad[4], need[4];
    #pragma unroll
                        for (uint32_t t = 0; t < 4u; ++t) {
                            const uint32_t w = t == 0 ? w0 : __builtin_amdgcn_alignbyte(w1, w0, t);   // bytes j+t, j+t+1, ...
                            const uint32_t L0 = (w >> 4) & 15u, b1 = (w >> 8) & 255u;
                            const bool ext = L0 == 15u;
                            d[t] = 3u + L0 + (ext ? b1 + 1u : 0u);                  // to the first byte after the offset
                            const uint32_t q = cstart + j + t + d[t];
                            bad[t] = (ext && b1 == 255u) || q >= fe;
                            need[t] = (w & 15u) == 15u;
                            qa[t] = bad[t] ? 0u : q - cstart;
                        }
                        lds_ld8x4(cbuf_a + qa[0], cbuf_a + qa[1], cbuf_a + qa[2], cbuf_a + qa[3], m1[0], m1[1], m1[2], m1[3]);
                        uint32_t o4 = 0;
    #pragma unroll
                        for (uint32_t t = 0; t < 4u; ++t) {
                            const uint32_t dd = d[t] + (need[t] ? 1u : 0u);
                            const bool b = bad[t] || (need[t] && m1[t] == 255u) || dd > 254u;
                            o4 |= (b ? 255u : dd) << (8u * t);
                        }
                        *reinterpret_cast<uint32_t*>(&nxt[j]) = o4;
                    }
                }
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
                uint32_t cutpos_w = 0;     // position of token #TOKCAP when a chunk has more tokens than the list holds
                // 4 input bytes at q (missing bytes past the end read as 0)
                auto rd4 = [&](uint32_t q) -> uint32_t {
                    const uint32_t r = q - cstart;
                    if (!STAGE) { if (q + 4u <= len) return ld4(in + q); }
                    else if (r + 4u <= kCB) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cbuf_a + r) : "memory"); return v; }
                    uint32_t v = 0;
                    for (uint32_t i = 0; i < 4u && q + i < len; ++i) v |= rdb(q + i) << (8u * i);
                    return v;
                };
                // One token at p (p < len): position of the next token; false on UnexpectedEnd.
                // decompress.rs:61-71 without the copies.  One LDS read covers the token and the first
                // length-extension byte, which is all a hop needs in the common cases.
                auto token_next = [&](uint32_t p, uint32_t& next) -> bool {
                    const uint32_t w = rd4(p);
                    const uint32_t tok = w & 255u;
                    uint32_t q = p + 1u;
                    uint32_t L = tok >> 4;
                    if (L == 15u) {
                        if (q >= len) return false;
                        uint32_t b = (w >> 8) & 255u; ++q;
                        L += b;
                        while (b == 255u) {
                            if (q >= len) return false;
                            b = rdb(q); ++q;
                            L += b; if (L > kMaxPosB) L = kMaxPosB;
                        }
                    }
                    if (len - q < L) return false;                    // :67 read_exact
                    q += L;
                    if (len - q < 2u) { next = len; return true; }    // :70 read_u16 fails: last literals
                    q += 2u;
                    if ((tok & 15u) == 15u) {
                        for (;;) {
                            if (q >= len) return false;
                            const uint32_t b = rdb(q); ++q;
                            if (b != 255u) break;
                        }
                    }
                    next = q;
                    return true;
                };
                // Walk the token chain from p up to (not including) the first token at or beyond `end`.
                // Tokens are counted in n and, when RECORD, their positions go to toks[k++].
                // Divergence control: a lane whose token needs more than the plain 4-byte view (length
                // extensions, end of input) parks; all other lanes keep hopping with one LDS read and a
                // handful of VALU per hop, and parked lanes are served together by the general routine.
                auto walk = [&](uint32_t p, const uint32_t end, uint32_t& n, uint32_t& k, bool& err, bool go, auto RECORD) -> uint32_t {
                    const uint32_t fast_end = len > 24u ? len - 24u : 0u;     // plain hops stay clear of the input's end
                    const uint32_t stop = end < fast_end ? end : fast_end;
                    uint32_t pclamp = len - 4u;                               // where idle lanes load from (only used when stop > 0)
                    if (STAGE && pclamp > cstart + kCB - 4u) pclamp = cstart + kCB - 4u;
                    for (;;) {
                        // Plain hops.  The scalar unit (one per CU) is the scarce issue resource of this kernel, so
                        // the loop is uniform: every lane executes every iteration with exec full, idle lanes are
                        // predicated with selects and load from a clamped address.  A lane goes idle when it is done
                        // or meets a token that needs more than the 4-byte view (0xFF runs, bodies leaving the staged
                        // bytes, the end of the input); such lanes are served below by the general routine.
                        if (!STAGE) {
                            uint32_t lim = go ? stop : 0u;
                            if (RECORD.value) hop_loop_record(p, lim, n, k, cutpos_w, in, fast_end, pclamp, cstart, lds_addr(toks), (uint32_t)TOKCAP,
                                                                     lds_addr(toks) + 4u * ((uint32_t)TOKCAP + lane));
                            else hop_loop(p, lim, n, in, fast_end, pclamp);
                        } else {
                            // nxt[] coordinates: position - cstart + address of nxt
                            const uint32_t base = nxt_a - cstart;
                            uint32_t pl = p + base, lim = go ? stop + base : 0u;
                            if (RECORD.value) {
                                uint32_t cut = 0;
                                thop_loop_record32(pl, lim, n, k, cut, nxt_a + kChunk - 1u, nxt_a, lds_addr(toks), (uint32_t)TOKCAP,
                                                 lds_addr(toks) + 4u * ((uint32_t)TOKCAP + lane));
                                if (cut) cutpos_w = cut - base;
                            } else thop_loop(pl, lim, n, nxt_a + kChunk - 1u);
                            p = pl - base;
                        }
                        // the general routine serves parked lanes and lanes near the end of the input
                        const bool slow = go && p < end && p < len;      // includes every lane that left the hop loop early
                        if (!__any(slow)) break;
                        if (slow) {
                            uint32_t nx;
                            if (!token_next(p, nx)) { err = true; p = len; }
                            else {
                                if (RECORD.value) { if (k < (uint32_t)TOKCAP) toks[k] = p - cstart; else if (k == (uint32_t)TOKCAP) cutpos_w = p; ++k; }
                                ++n; p = nx;
                            }
                        }
                    }
                    return p;
                };
                // =====================================================================
                // A. speculative lane-parallel parse of one chunk: regions [cstart + i*S, +S)
                // =====================================================================
                const uint32_t rbeg = cstart + lane * (uint32_t)S;
                const uint32_t rend = rbeg + (uint32_t)S;
                uint32_t start, x = 0, n = 0, kdummy = 0;
                bool lerr = false;
                if (STAGE) {
                    // A2. per region, exit of the token chain for every entry offset, by one backward sweep over nxt[]
                    // (a token is at least 3 bytes, so position p only depends on positions > p):
                    //   ex[p] = exit - rend (0..253) | 254: exits further away | 255: meets a token the table cannot express
                    const uint32_t exl = ex_a + lane * kExStride;
                    // per dword of nxt[]: positions 3, 2, 1 never depend on each other (a token is >= 3 bytes), position 0
                    // may depend on position 3 — two LDS round trips per four positions
    #pragma unroll 1
                    for (int wq = S / 4 - 1; wq >= 0; --wq) {
                        const uint32_t w4 = *reinterpret_cast<const uint32_t*>(&nxt[lane * (uint32_t)S + 4u * (uint32_t)wq]);
                        const uint32_t p0 = 4u * (uint32_t)wq;
                        uint32_t d[4], t[4], e[4];
    #pragma unroll
                        for (uint32_t u = 0; u < 4u; ++u) { d[u] = (w4 >> (8u * u)) & 255u; t[u] = p0 + u + d[u]; }
                        lds_ld8x4(exl + (t[3] < (uint32_t)S ? t[3] : p0), exl + (t[2] < (uint32_t)S ? t[2] : p0),
                                  exl + (t[1] < (uint32_t)S ? t[1] : p0), exl + p0, e[3], e[2], e[1], e[0]);
    #pragma unroll
                        for (uint32_t u = 3; u >= 1u; --u) {
                            if (t[u] >= (uint32_t)S) e[u] = t[u] - (uint32_t)S < 254u ? t[u] - (uint32_t)S : 254u;
                            if (d[u] == 255u) e[u] = 255u;
                            lds_st8(exl + p0 + u, e[u]);
                        }
                        e[0] = lds_ld8(exl + (t[0] < (uint32_t)S ? t[0] : p0));
                        if (t[0] >= (uint32_t)S) e[0] = t[0] - (uint32_t)S < 254u ? t[0] - (uint32_t)S : 254u;
                        if (d[0] == 255u) e[0] = 255u;
                        lds_st8(exl + p0, e[0]);
                    }
                    // A3. fixed point over the region starts with one table lookup per lane and pass:
                    //   start[i+1] = max(exit[0..i]); lane 0 starts at a true token, so the fixed point is the true chain.
                    start = lane == 0 ? cstart : rbeg;
                    bool redo = true;
                    for (uint32_t pass = 0; pass < 70u; ++pass) {
                        // branch-free: every lane looks its exit up every pass (same result for an unchanged start); only the rare
                        // entries the table cannot express (254 / 255) are walked, and only when the start has changed
                        const bool inreg = start < rend;                  // else the chain jumps over this region
                        const uint32_t e = lds_ld8(exl + (inreg ? start - rbeg : 0u));
                        const bool hc = inreg && e >= 254u;
                        const bool hard = hc && redo;
                        if (!hc) x = inreg ? rend + e : start;
                        if (__any(hard)) {                                // rare: walk it (0xFF runs, long literals, end of input)
                            uint32_t n1 = 0; bool e1 = false;
                            const uint32_t x1 = walk(start, rend, n1, kdummy, e1, hard, No{});
                            if (hard) x = x1;
                        }
                        const uint32_t nstart = wave_prev(wave_scan_max(x), cstart);
                        redo = nstart != start;
                        if (!__any(redo)) break;
                        start = nstart;
                    }
                    // A4. count the tokens of every region from its true start (also finds UnexpectedEnd)
                    x = walk(start, rend, n, kdummy, lerr, true, No{});
                } else {
                // First guess: walk in from the previous region's start (for lane 1 that is a true token),
                // so that the chain has usually re-synchronised by the time it enters the lane's region.
                start = lane == 0 ? cstart : rbeg - (uint32_t)S;
                {
                    uint32_t nw = 0; bool ew = false;
                    start = walk(start, rbeg, nw, kdummy, ew, true, No{});      // warm-up: these tokens do not count
                }
                bool redo = true;                          // lanes whose start changed walk again; the others keep x, n
                for (uint32_t pass = 0; pass < 70u; ++pass) {
                    if (redo) { n = 0; lerr = false; }
                    uint32_t n1 = 0; bool e1 = false;
                    const uint32_t x1 = walk(start, rend, n1, kdummy, e1, redo, No{});
                    if (redo) { x = x1; n = n1; lerr = e1; }
                    // true exits never decrease along the stream, so a lane starts at the largest exit
                    // before it (a long literal run hands its exit to every region it skips at once)
                    const uint32_t nstart = wave_prev(wave_scan_max(x), cstart);
                    redo = nstart != start;
                    if (!__any(redo)) break;               // this pass ran from the true starts
                    start = nstart;
                }
                }
                // token ranks in stream order
                const uint32_t incl_n = wave_scan_add(n);
                const uint32_t rank0 = incl_n - n;
                const uint32_t T = __builtin_amdgcn_readlane(incl_n, 63);
                const bool cut = T > (uint32_t)TOKCAP;
                const uint32_t Tc = cut ? (uint32_t)TOKCAP : T;
                {   // record pass
                    uint32_t k = rank0, n2 = 0; bool e2 = false;
                    (void)walk(start, rend, n2, k, e2, true, Yes{});
                }
                uint32_t cend_;       // where the next chunk starts
                int cerr = LZF_OK;    // UnexpectedEnd right after the listed tokens
                if (cut) {
                    cend_ = __builtin_amdgcn_readlane(cutpos_w, first_lane(__ballot(rank0 <= (uint32_t)TOKCAP && rank0 + n > (uint32_t)TOKCAP)) & 63u);
                } else {
                    cend_ = __builtin_amdgcn_readlane(x, 63);
                    if (__ballot(lerr)) cerr = LZF_UNEXPECTED_END;
                }

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
                    cend = cend_;
                }
                if (lane == 0u) ctl_valid[bsel] = valid ? 1 : 0;
                __syncthreads();                 // chunk kc is parsed (and the copier is done with chunk kc - 1)
                if (!valid) break;
                cstart = cend;
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

#define LZF_INSTP(NAME, RG, S_, T) template __global__ void lzf_decompress_paired_kernel<RG, S_, T>(const lzf_decompress_job*, lzf_job_result*, uint32_t);
LZF_PAIRED_VARIANTS(LZF_INSTP)
#undef LZF_INSTP

}  // namespace lzf
