// lz4_decompress_batched.hip — batched raw::decompress_raw for gfx950, second generation.
//
// Same contract as lz4_decompress.hip (src/raw/decompress.rs:58-138, one wavefront per block).
// The sequential token walk of LZ4 decoding is the part a GPU does worst (one dependent,
// data-dependent hop per sequence), so this kernel splits decoding into two stages per chunk
// of compressed input:
//
//   PARSE (lane-parallel, speculative).  A chunk is 64 regions of S bytes, one per lane.  The exit of
//   a region (first token at or beyond its end) is the true start of the next one, so the wave
//   iterates  start[i] <- max(exit[0..i-1])  until nothing changes; lane 0 starts from a known-true
//   token, hence the fixed point is the true chain (lane i is exact after i+1 passes at worst).
//   How a lane gets its region's exit depends on the variant:
//     STAGE = true  (default, S = 16): the chunk is staged in LDS with one coalesced 16 B/lane pass;
//       nxt[] = distance to the next token for EVERY byte position (all lanes busy, no dependent chain),
//       ex[] = per region, exit for every entry offset (one backward sweep over nxt[]); a pass of the
//       fixed point is then one table lookup per lane.  Counting and recording the tokens is a
//       hand-scheduled hop loop over nxt[] (lzf_parse_helpers.h).  Every later access to compressed
//       bytes (token re-read, offsets, literals) is an LDS access too.
//     STAGE = false (direct4w, S = 256): a lane walks its region hop by hop reading tokens from
//       HBM/L2 (hand-scheduled loop as well); first guesses come from a warm-up walk.
//   A final pass records the token positions, compacted in stream order, into an LDS list.
//
//   COPY (lane-parallel, batched).  64 consecutive sequences go to the 64 lanes.  The most
//   recent RING bytes of output live in an LDS ring (ring index == output address mod RING, so
//   16-byte chunks of the ring line up with 16-byte chunks of HBM).  A lane moves its literal run
//   and its match with "two-ended" pieces (first and last 8/4/2 bytes, any alignment: gfx950
//   executes misaligned DS accesses), so a run of up to 32 bytes costs at most 4 LDS reads and
//   4 LDS writes.  "far" matches (older than the ring's intact history) are read back from HBM;
//   "near" matches copy ring -> ring in rounds: every match whose source ends below the start of the
//   first unresolved match moves at once.  Overlapping matches use dst[t] = src[t mod offset], whose
//   reads all precede the match.  LDS executes one wave's accesses in order, so no barrier or wait
//   separates dependent copies.  The finished batch is flushed ring -> HBM with aligned
//   16-byte stores: every output byte is written to HBM exactly once, coalesced.
//   A sequence too large for a batch (> RING/3 output bytes) takes a solo path (cooperative
//   HBM -> HBM copies, then the ring is re-filled from HBM).
//
// Error precedence is the reference's: within a sequence literal EOF / LSIC EOF
// (UnexpectedEnd), MemoryLimitExceeded, ZeroDeduplicationOffset, InvalidDeduplicationOffset
// (decompress.rs:63-75,82-89); across sequences the first one in stream order wins.
#include "lzf_device.h"
#include "kernels.h"
#include "lzf_copy_helpers.h"
#include "lzf_parse_helpers.h"

namespace lzf {


template <int RING, int S, int TOKCAP, bool STAGE>
__global__ __launch_bounds__(64) void lzf_decompress_batched_kernel(
    const lzf_decompress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs,
    const uint32_t* __restrict__ perm) {
    constexpr uint32_t kMask = RING - 1;
#ifndef LZF_SPAN_DIV
#define LZF_SPAN_DIV 3
#endif
    constexpr uint32_t kSpanMax = RING / LZF_SPAN_DIV; // output bytes one batch may produce
    constexpr uint32_t kNearHist = RING - kSpanMax;    // history before the batch that stays intact in the ring
    constexpr uint32_t kChunk = 64u * S;               // compressed bytes whose tokens one parse covers
    constexpr uint32_t kCB = STAGE ? kChunk + 64u : 0u;   // staged bytes: the chunk + room for token bodies (0: read HBM/L2 directly)
    static_assert(kChunk <= 65536, "token positions are stored as u16 offsets into the chunk");
    static_assert(kCB % 16 == 0, "chunk buffer is filled in 16-byte pieces");
    constexpr uint32_t kCBAlloc = STAGE ? kCB : 16u;
    __shared__ __attribute__((aligned(16))) uint8_t ring[RING];
    __shared__ __attribute__((aligned(16))) uint8_t cbuf[kCBAlloc];
    __shared__ __attribute__((aligned(16))) uint8_t nxt[STAGE ? kChunk : 16u];   // staged variants: distance to the next token per position
    // toks: token list (+ one dump slot per lane for predicated stores).  Staged variants first use the same
    // bytes for ex[]: per region a table "entry offset -> exit", padded to S + 4 bytes per lane so that the
    // 64 lanes hit 64 different banks when they all touch the same offset.
    constexpr uint32_t kExStride = (uint32_t)S + 4u;
    constexpr uint32_t kTokBytes = ((uint32_t)TOKCAP + 64u) * 2u;
    constexpr uint32_t kExBytes = STAGE ? 64u * kExStride : 0u;
    __shared__ __attribute__((aligned(16))) uint8_t tokex[kTokBytes > kExBytes ? kTokBytes : kExBytes];
    uint16_t* const toks = reinterpret_cast<uint16_t*>(tokex);

    if (blockIdx.x >= n_jobs) return;
    const uint32_t jid = perm ? perm[blockIdx.x] : blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const lzf_decompress_job job = jobs[jid];
    const long long t_start = clock64();

    int status = LZF_OK;
    uint32_t o = 0;
#ifdef LZF_PHASE_TIMING
    long long g_tph[6] = {0, 0, 0, 0, 0, 0}; uint32_t g_pk_hi = 0;
#endif
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
        const uint32_t rb = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u);   // ring bias
        const uint32_t ring_a = lds_addr(ring), cbuf_a = lds_addr(cbuf), nxt_a = lds_addr(nxt), ex_a = lds_addr(tokex);
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

        o = (uint32_t)job.out_existing_len;
        uint32_t safe = o;   // out[0, safe) is visible to this wave's global loads
        if (o > 0) ring_fill(o > (uint32_t)RING ? o - RING : 0u, o);   // Vec content on entry = history

#ifdef LZF_PHASE_TIMING
        long long tph[6] = {0, 0, 0, 0, 0, 0}; long long tq = clock64();
#define PHASE(i) do { const long long tn = clock64(); tph[i] += tn - tq; tq = tn; } while (0)
#else
#define PHASE(i) do { } while (0)
#endif
        uint32_t cstart = 0;                 // a true token position (or len)
        while (cstart < len && status == LZF_OK) {
#define LZF_TOK_T uint16_t
#define LZF_THOP_RECORD thop_loop_record
#include "lz4_decompress_parse_phase.inc"
#undef LZF_THOP_RECORD
#undef LZF_TOK_T
            PHASE(0);

#define LZF_TOKEN_AT(i) toks[(i)]
#include "lz4_decompress_batch_phase.inc"
#undef LZF_TOKEN_AT
            if (status == LZF_OK && cerr != LZF_OK) status = cerr;
            cstart = cend;
        }
#ifdef LZF_PHASE_TIMING
        for (int i = 0; i < 6; ++i) g_tph[i] = tph[i];
#endif
#undef RIDX
    }
    if (lane == 0) {
        results[jid].out_len = o;
#ifdef LZF_PHASE_TIMING
        // debug build only: six phase totals, 10 bits each in units of 2^20 cycles, above bit 32 / in `reserved`
        {
            unsigned long long pk = 0;
            for (int i = 0; i < 6; ++i) { unsigned long long u = (unsigned long long)(g_tph[i] >> 20); if (u > 1023) u = 1023; pk |= u << (10 * i); }
            results[jid].out_len = (unsigned long long)o | ((pk & 0xFFFFFFFFull) << 32);
            g_pk_hi = (uint32_t)(pk >> 32);
        }
#endif
        results[jid].status = status;
#ifdef LZF_PHASE_TIMING
        results[jid].reserved = g_pk_hi;
#else
        results[jid].reserved = (uint32_t)((clock64() - t_start) >> 10);   // diagnostic: shader kilo-cycles spent on this job
#endif
    }
}

#define LZF_INST(NAME, R, S_, T, ST) template __global__ void lzf_decompress_batched_kernel<R, S_, T, ST>(const lzf_decompress_job*, lzf_job_result*, uint32_t, const uint32_t*);
LZF_DECOMPRESS_VARIANTS(LZF_INST)
#undef LZF_INST

}  // namespace lzf
