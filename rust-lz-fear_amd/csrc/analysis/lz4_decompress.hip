// lz4_decompress.hip — batched raw::decompress_raw for gfx950 (MI355X), wave64.
//
// Replaces src/raw/decompress.rs:58-138 of lz-fear (decompress_raw, read_lsic,
// copy_overlapping) for many blocks per launch.  One wavefront decodes one block:
//   * the token / LSIC / offset parse is wave-uniform (SGPR state, scalar branches).  Parse
//     bytes come from a 256-byte register window over the compressed input (lane i holds
//     dword i, the next window is prefetched), read with v_readlane — no memory latency on
//     the token chain;
//   * literal and match copies are cooperative: 16 B per lane where the run allows it,
//     one byte per lane below that; overlapping matches (offset < length) use the
//     period-`offset` formulation dst[i] = hist[i mod offset], whose sources all precede
//     the match, so every lane can run at once;
//   * output written by this wave is re-read (match sources) through global memory; a
//     per-wave `safe` watermark tracks which output bytes are known to have landed and a
//     store fence (s_waitcnt vmcnt(0)) is issued only when a match reaches above it.
//
// Error precedence is the reference's (decompress.rs:63-75,82-89): literal EOF, LSIC EOF ->
// UnexpectedEnd; then MemoryLimitExceeded; ZeroDeduplicationOffset; InvalidDeduplicationOffset.
#include "../lzf_device.h"

namespace lzf {

// 256-byte register window over the compressed input.
struct InWindow {
    cgu8* in;
    uint32_t len;
    uint32_t base;   // wave-uniform; 0xFFFFFFFF = nothing loaded
    uint32_t w;      // lane i: bytes [base+4i, base+4i+4)
    uint32_t wn;     // same for base+256 (prefetch)

    __device__ __forceinline__ uint32_t fetch(uint32_t b, uint32_t lane) const {
        const uint32_t a = b + lane * 4u;
        uint32_t v = 0;
        if (a + 4u <= len) {
            v = ld4(in + a);
        } else if (a < len) {  // last, partial dword of the block: never read past input_len
            for (uint32_t i = 0; a + i < len; ++i) v |= (uint32_t)in[a + i] << (8u * i);
        }
        return v;
    }
    // byte at p (p < len, wave-uniform)
    __device__ __forceinline__ uint32_t byte(uint32_t p, uint32_t lane) {
        const uint32_t b = p & ~255u;
        if (b != base) {
            if (b == base + 256u) w = wn; else w = fetch(b, lane);
            base = b;
            if (b + 256u < len) wn = fetch(b + 256u, lane);
        }
        const uint32_t d = __builtin_amdgcn_readlane(w, (p >> 2) & 63u);
        return (d >> ((p & 3u) * 8u)) & 0xFFu;
    }
};

constexpr uint32_t kMaxPos = 0x7FFFFF00u;   // positions are 32-bit inside the kernel

__global__ __launch_bounds__(64) void lzf_decompress_wave_kernel(
    const lzf_decompress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs,
    const uint32_t* __restrict__ perm) {
    if (blockIdx.x >= n_jobs) return;
    const uint32_t jid = perm ? perm[blockIdx.x] : blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const lzf_decompress_job job = jobs[jid];
    const long long t_start = clock64();

    int status = LZF_OK;
    uint32_t o = 0;
    if (job.input_len >= kMaxPos || job.out_existing_len >= kMaxPos || job.prefix_len >= kMaxPos || job.out_existing_len > job.out_cap) {
        status = LZF_CONTRACT;   // blocks beyond 2 GiB are outside this kernel's contract
    } else {
        cgu8* __restrict__ in = as_global(job.input);
        cgu8* __restrict__ prefix = as_global(job.prefix);
        gu8* out = as_global(job.out);
        const uint32_t len = (uint32_t)job.input_len;
        const uint32_t plen = (uint32_t)job.prefix_len;
        const uint32_t cap = job.out_cap > kMaxPos ? kMaxPos : (uint32_t)job.out_cap;
        const uint64_t limit = job.output_limit;
        o = (uint32_t)job.out_existing_len;
        uint32_t safe = o;   // out[0, safe) is known to be visible to this wave's loads
        uint32_t p = 0;
        InWindow win{in, len, 0xFFFFFFFFu, 0u, 0u};

        while (p < len) {                                              // decompress.rs:61
            const uint32_t token = win.byte(p, lane); ++p;
            // ---- literal length: read_lsic, decompress.rs:30-43,63
            uint32_t lit = token >> 4;
            if (lit == 15u) {
                bool eof = false;
                for (;;) {
                    if (p >= len) { eof = true; break; }
                    const uint32_t b = win.byte(p, lane); ++p;
                    lit = lit + b; if (lit > kMaxPos) lit = kMaxPos;
                    if (b != 255u) break;
                }
                if (eof) { status = LZF_UNEXPECTED_END; break; }
            }
            if (len - p < lit) { status = LZF_UNEXPECTED_END; break; }  // :67 read_exact
            if (cap - o < lit) { status = LZF_OUT_CAPACITY; break; }
            wave_copy(out + o, in + p, lit, lane);                      // :65-67 (no limit check)
            p += lit; o += lit;

            if (len - p < 2u) break;                                    // :70 read_u16 Err: stop
            uint32_t offset = win.byte(p, lane); offset |= win.byte(p + 1u, lane) << 8; p += 2u;
            // ---- match length: 4 + read_lsic, :71
            uint32_t mlen = token & 15u;
            if (mlen == 15u) {
                bool eof = false;
                for (;;) {
                    if (p >= len) { eof = true; break; }
                    const uint32_t b = win.byte(p, lane); ++p;
                    mlen = mlen + b; if (mlen > kMaxPos) mlen = kMaxPos;
                    if (b != 255u) break;
                }
                if (eof) { status = LZF_UNEXPECTED_END; break; }
            }
            mlen += 4u;
            if ((uint64_t)o + mlen > limit) { status = LZF_MEMORY_LIMIT_EXCEEDED; break; }   // :72-74
            if (offset == 0u) { status = LZF_ZERO_DEDUP_OFFSET; break; }                      // :83
            if (offset > o) {                                                                 // :84-99
                const uint32_t need = offset - o;
                if (need > plen) { status = LZF_INVALID_DEDUP_OFFSET; break; }                // :87-89
                const uint32_t n = need < mlen ? need : mlen;                                 // :90
                if (cap - o < n) { status = LZF_OUT_CAPACITY; break; }
                wave_copy(out + o, prefix + (plen - need), n, lane);
                o += n; mlen -= n;          // rest comes from out[0..): offset now equals o
                if (mlen == 0u) continue;
            }
            if (cap - o < mlen) { status = LZF_OUT_CAPACITY; break; }
            // ---- copy_overlapping :100-135, as a period-`offset` parallel copy
            const uint32_t src0 = o - offset;
            const uint32_t span = mlen < offset ? mlen : offset;   // distinct source bytes
            if (src0 + span > safe) { wave_store_fence(); safe = o; }
            cgu8* hist = out + src0;
            gu8* dst = out + o;
            if (mlen <= offset) {
                wave_copy(dst, hist, mlen, lane);                   // :104-111 non-overlapping
            } else if (offset == 1u) {                              // :102 memset
                const uint32_t b = hist[0];
                const uint32_t b4 = b * 0x01010101u;
                const u32x4 v = {b4, b4, b4, b4};
                const uint32_t bulk = mlen & ~15u;
                for (uint32_t i = lane * 16u; i < bulk; i += kWave * 16u) st16(dst + i, v);
                if (lane < mlen - bulk) dst[bulk + lane] = (uint8_t)b;
            } else {                                                // :112-135 overlapping
                uint32_t r = lane % offset;
                const uint32_t adv = kWave % offset;
                for (uint32_t i = lane; i < mlen; i += kWave) {
                    dst[i] = hist[r];
                    r += adv; if (r >= offset) r -= offset;
                }
            }
            o += mlen;
        }
    }
    if (lane == 0) {
        results[jid].out_len = o;
        results[jid].status = status;
        results[jid].reserved = (uint32_t)((clock64() - t_start) >> 10);   // diagnostic: shader kilo-cycles spent on this job
    }
}

}  // namespace lzf
