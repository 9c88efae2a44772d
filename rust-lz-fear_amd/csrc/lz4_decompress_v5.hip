// lz4_decompress_v5.hip — raw::decompress_raw (src/raw/decompress.rs:58-138) for gfx950, fifth generation.
//
// A producer / consumer pair of wavefronts per block: wave 0 PARSES chunk k+1 of the compressed input while wave 1 COPIES
// chunk k; the waves meet at one barrier per chunk.  What changed against the earlier pairs (lz4_decompress_paired.hip):
//   * the parser walks LARGE regions (64 lanes x S = 512 bytes per chunk) straight out of HBM/L2 — at that size a walk from an
//     arbitrary byte re-synchronises with the true token chain inside its region almost always, so the fixed point over the
//     region starts takes ~3 cheap passes instead of ~20, and nothing of the compressed input is staged in LDS
//     (lz4_decompress_gwalk_phase.inc);
//   * the chunk's token list (32-bit entries) goes through a double-buffered list in global scratch (L2-resident: written by
//     the parser wave, read once by the copier wave with L1-bypassing loads);
//   * LDS holds only the copier's output window and the parser's mark bits, so a CU keeps many more pairs in flight;
//   * the copy stage is lz4_decompress_copy3.inc: linear window, matches of a batch copied at once and repeated until stable.
// Error precedence is the reference's: within a sequence literal EOF / LSIC EOF (UnexpectedEnd), MemoryLimitExceeded,
// ZeroDeduplicationOffset, InvalidDeduplicationOffset (decompress.rs:63-75,82-89); across sequences the first in stream order.
#include "lzf_device.h"
#include "kernels.h"
#include "lzf_copy_helpers.h"
#include "lzf_parse_helpers.h"
#include <type_traits>

namespace lzf {

template <int W, int S, bool STAGED>
__global__ __launch_bounds__(128) void lzf_decompress_v5_kernel(
    const lzf_decompress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs,
    const uint32_t* __restrict__ perm, uint32_t* __restrict__ scratch, uint32_t base) {
    constexpr int SPAN = 3 * W / 8;                    // output bytes one batch may produce
    constexpr int HKEEP = W / 2;                       // history a slide keeps
    static_assert(HKEEP + SPAN + 96 <= W && W % 1024 == 0, "a batch fits behind the kept history");
    constexpr uint32_t kChunk = 64u * S;               // compressed bytes whose tokens one parse covers
    static_assert(kChunk <= 65536 && S % 128 == 0, "token positions are 16-bit chunk offsets; mark rows are cleared 16 bytes at a time");
    constexpr int TOKCAP = LZF_V5_TOKCAP(S);           // a token is at least 3 bytes
    __shared__ __attribute__((aligned(16))) uint8_t win[W + 32 + 512 + 16 + 1024 + 32]; // window + per-lane scratch words + the batch's 1 KB of compressed input
    __shared__ __attribute__((aligned(16))) uint8_t marks[kChunk / 8u];             // parser: visited positions of pass 0, one row per lane
    constexpr uint32_t kCB = kChunk + 64u;             // staged bytes: the chunk + room for token bodies
    __shared__ __attribute__((aligned(16))) uint8_t cbufs[STAGED ? 16u + kCB + 16u : 16u];   // parser: the chunk's bytes (staged variants)
    __shared__ uint32_t ctl_T[2], ctl_cstart[2];
    __shared__ int ctl_err[2], ctl_valid[2], ctl_stop;

    if (base + blockIdx.x >= n_jobs) return;
    const uint32_t jid = perm ? perm[base + blockIdx.x] : base + blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0: parser, 1: copier (uniform per wavefront)
    const lzf_decompress_job job = jobs[jid];
    const long long t_start = clock64();
    LZF_GLOBAL uint32_t* const glist = (LZF_GLOBAL uint32_t*)scratch + (size_t)blockIdx.x * (2u * LZF_V5_LISTWORDS(S));

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

        if (threadIdx.x == 0) ctl_stop = 0;
        __syncthreads();
        if (role == 0u) {
            // ================================ PARSER ================================
            const uint32_t marks_a = lds_addr(marks);
            uint32_t cstart = 0;                 // a true token position (or len)
            for (uint32_t kc = 0;; ++kc) {
                const uint32_t bsel = kc & 1u;
                LZF_GLOBAL uint32_t* const gtoks = glist + bsel * LZF_V5_LISTWORDS(S);
                const bool valid = cstart < len && *(volatile int*)&ctl_stop == 0;
                uint32_t cend_next = cstart;
                if (valid) {
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
                    if (lane == 0u) { ctl_T[bsel] = Tc; ctl_cstart[bsel] = cstart; ctl_err[bsel] = cerr; }
                    cend_next = cend;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the list's stores have left this wave (L2 holds them)
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
            for (uint32_t kc = 0;; ++kc) {
                __syncthreads();                 // chunk kc is parsed
                const uint32_t bsel = kc & 1u;
                if (*(volatile int*)&ctl_valid[bsel] == 0) break;
                if (status != LZF_OK) continue;  // keep meeting the parser until it sees the stop flag
                LZF_GLOBAL uint32_t* const gtoks = glist + bsel * LZF_V5_LISTWORDS(S);
                const uint32_t cstart = *(volatile uint32_t*)&ctl_cstart[bsel];
                const uint32_t Tc = *(volatile uint32_t*)&ctl_T[bsel];
                const int cerr = *(volatile int*)&ctl_err[bsel];
                // (L1-bypassing load: the list was written by the other wave of this workgroup, and this buffer has been used before)
#define LZF_TOKEN_WORD(i) __hip_atomic_load(gtoks + (i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#include "lz4_decompress_copy3.inc"
#undef LZF_TOKEN_WORD
                if (status == LZF_OK && cerr != LZF_OK) status = cerr;
                if (status != LZF_OK && lane == 0u) *(volatile int*)&ctl_stop = 1;
            }
            win_flush(fl, o);                    // the last partial granule
        }
    }
    if (role == 1u && lane == 0u) {
        results[jid].out_len = o;
        results[jid].status = status;
        results[jid].reserved = (uint32_t)((clock64() - t_start) >> 10);   // diagnostic: shader kilo-cycles spent on this job
    }
}

#define LZF_INST5(NAME, W_, S_, ST) template __global__ void lzf_decompress_v5_kernel<W_, S_, ST>(const lzf_decompress_job*, lzf_job_result*, uint32_t, const uint32_t*, uint32_t*, uint32_t);
LZF_V5_VARIANTS(LZF_INST5)
#undef LZF_INST5

}  // namespace lzf
