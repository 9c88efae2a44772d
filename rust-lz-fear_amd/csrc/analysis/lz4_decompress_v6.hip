// lz4_decompress_v6.hip — raw::decompress_raw (src/raw/decompress.rs:58-138) for gfx950 as TWO kernels over a batch of blocks.
//
//   lzf_v6_plan_kernel    one workgroup: where each block's token list and chunk table start in the scratch area
//                         (prefix sums over the jobs' compressed sizes: a token is at least 3 bytes).
//   lzf_v6_parse_kernel   one wavefront per block: the region-walk parse of lz4_decompress_gwalk_phase.inc, chunk after
//                         chunk (64 lanes x S bytes), straight out of HBM/L2.  It needs no LDS beyond its mark bits and few
//                         registers, so a CU holds 32 of these waves and their dependent loads overlap.  Output: the block's
//                         token list (32-bit entries: chunk offset | L << 16 | (M - 4) << 24) and one {first token, first byte}
//                         pair per chunk.
//   lzf_v6_copy_kernel    one wavefront per block: lz4_decompress_copy3.inc over the listed tokens (linear LDS window, batches
//                         of 64 sequences, token words / compressed bytes prefetched one batch ahead).
// Splitting the pair of (retired) v5 pair kernel in two launches doubles the copy waves a CU holds (they are the critical path)
// and lets each kernel have its own register budget; the price is the token lists in HBM (4 bytes per sequence written and
// read once: + ~30 % traffic on top of the compressed and decoded bytes).
#include "../lzf_device.h"
#include "../kernels.h"
#include "../lzf_copy_helpers.h"
#include "../lzf_parse_helpers.h"
#include <type_traits>

namespace lzf {

// per block: [0] chunks, [1] tokens, [2] status of the parse (UnexpectedEnd right after the listed tokens), [3] unused;
// then 2 words per chunk: first token index, first byte
__global__ __launch_bounds__(1024) void lzf_v6_plan_kernel(const lzf_decompress_job* __restrict__ jobs, uint32_t n_jobs, uint32_t base, uint32_t stride, uint32_t cnt,
                                                          const uint32_t* __restrict__ perm, uint32_t chunk_bytes,
                                                          uint64_t* __restrict__ tok_off, uint64_t* __restrict__ tab_off) {
    // exclusive prefix sums over the slice [base, base + cnt) in launch order; entry cnt = totals
    __shared__ uint64_t s_tok[1024], s_tab[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (cnt + 1023u) / 1024u;
    uint64_t a = 0, b = 0;
    for (uint32_t i = t * per; i < (t + 1u) * per && i < cnt; ++i) {
        const uint32_t j = perm ? perm[base + i * stride] : base + i * stride;
        const uint64_t len = jobs[j].input_len < 0x7FFFFF00ull ? jobs[j].input_len : 0;
        a += len / 3u + 64u;
        b += 4u + 2u * (len / chunk_bytes + 2u);
    }
    s_tok[t] = a; s_tab[t] = b;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint64_t x = t >= d ? s_tok[t - d] : 0, y = t >= d ? s_tab[t - d] : 0;
        __syncthreads();
        s_tok[t] += x; s_tab[t] += y;
        __syncthreads();
    }
    uint64_t ea = s_tok[t] - a, eb = s_tab[t] - b;
    for (uint32_t i = t * per; i < (t + 1u) * per && i < cnt; ++i) {
        const uint32_t j = perm ? perm[base + i * stride] : base + i * stride;
        const uint64_t len = jobs[j].input_len < 0x7FFFFF00ull ? jobs[j].input_len : 0;
        tok_off[i] = ea; tab_off[i] = eb;
        ea += len / 3u + 64u;
        eb += 4u + 2u * (len / chunk_bytes + 2u);
    }
    if (t == 1023u) { tok_off[cnt] = s_tok[t]; tab_off[cnt] = s_tab[t]; }
    (void)n_jobs;
}

template <int S, bool STAGED>
__global__ __launch_bounds__(64) void lzf_v6_parse_kernel(const lzf_decompress_job* __restrict__ jobs, uint32_t n_jobs, uint32_t base, uint32_t stride,
                                                          const uint32_t* __restrict__ perm, const uint64_t* __restrict__ tok_off,
                                                          const uint64_t* __restrict__ tab_off, uint32_t* __restrict__ toks_all, uint32_t* __restrict__ tabs_all) {
    constexpr uint32_t kChunk = 64u * S;
    static_assert(kChunk <= 65536 && S % 128 == 0, "token positions are 16-bit chunk offsets; mark rows are cleared 16 bytes at a time");
    constexpr int TOKCAP = (64 * S) / 3 + 1;           // a token is at least 3 bytes: a chunk never has more
    __shared__ __attribute__((aligned(16))) uint8_t marks[kChunk / 8u];
    constexpr uint32_t kCB = kChunk + 64u;             // staged bytes: the chunk + room for token bodies
    __shared__ __attribute__((aligned(16))) uint8_t cbufs[STAGED ? 16u + kCB + 16u : 16u];
    if (base + blockIdx.x * stride >= n_jobs) return;
    const uint32_t jid = perm ? perm[base + blockIdx.x * stride] : base + blockIdx.x * stride;
    const uint32_t lane = threadIdx.x & 63u;
    const lzf_decompress_job job = jobs[jid];
    LZF_GLOBAL uint32_t* const list = (LZF_GLOBAL uint32_t*)toks_all + tok_off[blockIdx.x];
    LZF_GLOBAL uint32_t* const tab = (LZF_GLOBAL uint32_t*)tabs_all + tab_off[blockIdx.x];
    if (job.input_len >= kMaxPosB) { if (lane == 0u) { tab[0] = 0; tab[1] = 0; tab[2] = LZF_CONTRACT; } return; }
    cgu8* __restrict__ in = as_global(job.input);
    const uint32_t len = (uint32_t)job.input_len;
    const uint32_t marks_a = lds_addr(marks);
    uint32_t cstart = 0, tbase = 0, nchunks = 0;
    int perr = LZF_OK;
    while (cstart < len && perr == LZF_OK) {
        LZF_GLOBAL uint32_t* const gtoks = list + tbase;
        uint32_t Tc_, cend_; int cerr_;
        if constexpr (STAGED) {
            // ---- stage in[cstart, cstart + kCB) in LDS (zeros beyond the input): every hop is then one LDS read
            const uint32_t cbuf_a = lds_addr(cbufs) + 16u;
            {
                uint8_t* const cbuf = cbufs + 16u;
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
                        else if (i < avail) { for (uint32_t t = 0; i + t < avail; ++t) v[k][(t >> 2) & 3u] |= (uint32_t)g[i + t] << ((t & 3u) * 8u); }
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) {
                        const uint32_t i = b4 + k * 1024u + lane * 16u;
                        if (i < kCB) *reinterpret_cast<u32x4*>(&cbuf[i]) = v[k];
                    }
                }
            }
#define LZF_GWALK_STAGED 1
#include "lz4_decompress_gwalk_phase.inc"
#undef LZF_GWALK_STAGED
            Tc_ = Tc; cend_ = cend; cerr_ = cerr;
        } else {
#include "lz4_decompress_gwalk_phase.inc"
            Tc_ = Tc; cend_ = cend; cerr_ = cerr;
        }
        const uint32_t Tc = Tc_, cend = cend_; const int cerr = cerr_;
        if (lane == 0u) { tab[4u + 2u * nchunks] = tbase; tab[5u + 2u * nchunks] = cstart; }
        ++nchunks; tbase += Tc;
        perr = cerr;
        cstart = cend;
    }
    if (lane == 0u) { tab[0] = nchunks; tab[1] = tbase; tab[2] = (uint32_t)perr; tab[4u + 2u * nchunks] = tbase; tab[5u + 2u * nchunks] = cstart; }
}

template <int W>
__global__ __launch_bounds__(64) void lzf_v6_copy_kernel(const lzf_decompress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs,
                                                         uint32_t base, uint32_t stride, const uint32_t* __restrict__ perm, const uint64_t* __restrict__ tok_off,
                                                         const uint64_t* __restrict__ tab_off, const uint32_t* __restrict__ toks_all,
                                                         const uint32_t* __restrict__ tabs_all) {
    constexpr int SPAN = 3 * W / 8;                    // output bytes one batch may produce
    constexpr int HKEEP = W / 2;                       // history a slide keeps
    static_assert(HKEEP + SPAN + 96 <= W && W % 1024 == 0, "a batch fits behind the kept history");
    __shared__ __attribute__((aligned(16))) uint8_t win[W + 32 + 512 + 16 + 1024 + 32]; // window + per-lane scratch words + the batch's 1 KB of compressed input
    if (base + blockIdx.x * stride >= n_jobs) return;
    const uint32_t jid = perm ? perm[base + blockIdx.x * stride] : base + blockIdx.x * stride;
    const uint32_t lane = threadIdx.x & 63u;
    const lzf_decompress_job job = jobs[jid];
    const long long t_start = clock64();
    const LZF_GLOBAL uint32_t* const list = (const LZF_GLOBAL uint32_t*)toks_all + tok_off[blockIdx.x];
    const LZF_GLOBAL uint32_t* const tab = (const LZF_GLOBAL uint32_t*)tabs_all + tab_off[blockIdx.x];

    int status = LZF_OK;
    uint32_t o = 0;
    if (job.input_len >= kMaxPosB || job.out_existing_len >= kMaxPosB || job.prefix_len >= kMaxPosB || job.out_existing_len > job.out_cap) {
        status = LZF_CONTRACT;
    } else {
        cgu8* __restrict__ in = as_global(job.input);
        cgu8* __restrict__ prefix = as_global(job.prefix);
        gu8* out = as_global(job.out);
        const uint32_t len = (uint32_t)job.input_len;
        const uint32_t plen = (uint32_t)job.prefix_len;
        const uint32_t cap = job.out_cap > kMaxPosB ? kMaxPosB : (uint32_t)job.out_cap;
        const uint64_t limit = job.output_limit;
        const uint32_t rb = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u);   // out + x is 16-byte aligned when (x + rb) % 16 == 0
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
        long long dbg_cycles = 0, dbg_t0 = 0; int dbg_sec = 0; (void)dbg_cycles; (void)dbg_t0; (void)dbg_sec;
        hlo = o > (uint32_t)HKEEP ? o - (uint32_t)HKEEP : 0u;
        wlo = AL(hlo);
        if (o > hlo) win_fill(hlo, o);       // Vec content on entry = history
        fl = o;
        auto rdb = [&](uint32_t q) -> uint32_t { return (uint32_t)in[q]; };
        // 4 input bytes at q (missing bytes past the end read as 0)
        auto rd4 = [&](uint32_t q) -> uint32_t {
            if (q + 4u <= len) return ld4(in + q);
            uint32_t v = 0;
            for (uint32_t i = 0; i < 4u && q + i < len; ++i) v |= (uint32_t)in[q + i] << (8u * i);
            return v;
        };
        const uint32_t nchunks_all = tab[0];
        const int perr = (int)tab[2];
        if (perr == LZF_CONTRACT) status = LZF_CONTRACT;
        for (uint32_t kc = 0; kc < nchunks_all && status == LZF_OK; ++kc) {
            const uint32_t t0c = tab[4u + 2u * kc], cstart = tab[5u + 2u * kc];
            const uint32_t Tc = tab[6u + 2u * kc] - t0c;
            const LZF_GLOBAL uint32_t* const gtoks = list + t0c;
#define LZF_TOKEN_WORD(i) gtoks[(i)]
#include "lz4_decompress_copy3.inc"
#undef LZF_TOKEN_WORD
        }
        if (status == LZF_OK && perr != LZF_OK) status = perr;
        win_flush(fl, o);                    // the last partial granule
        if (LZF_DBG_TIME && lane == 0u) results[jid].reserved = (uint32_t)(dbg_cycles >> 10);
    }
    if (lane == 0u) {
        results[jid].out_len = o;
        results[jid].status = status;
        if (!LZF_DBG_TIME) results[jid].reserved = (uint32_t)((clock64() - t_start) >> 10);   // diagnostic: shader kilo-cycles spent on this job
    }
}

#define LZF_INST6P(S_, ST) template __global__ void lzf_v6_parse_kernel<S_, ST>(const lzf_decompress_job*, uint32_t, uint32_t, uint32_t, const uint32_t*, const uint64_t*, const uint64_t*, uint32_t*, uint32_t*);
LZF_INST6P(512, false)
LZF_INST6P(256, true)
LZF_INST6P(128, true)
LZF_INST6P(384, true)
#undef LZF_INST6P
#define LZF_INST6C(W_) template __global__ void lzf_v6_copy_kernel<W_>(const lzf_decompress_job*, lzf_job_result*, uint32_t, uint32_t, uint32_t, const uint32_t*, const uint64_t*, const uint64_t*, const uint32_t*, const uint32_t*);
LZF_INST6C(4096)
LZF_INST6C(6144)
#undef LZF_INST6C

}  // namespace lzf
