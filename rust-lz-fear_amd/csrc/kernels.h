// kernels.h — declarations of the gfx950 kernels for the C-ABI translation unit.
//
// The product library (build.py, no defines) holds the kernels lzf_decompress_batch / lzf_compress_batch launch:
// paired48, paired24, staged16, the two compress kernels and the small helpers.  -DLZF_ANALYSIS (liblzfear_hip_analysis.so)
// adds the first-generation kernel and the tuning variants of the batched / paired kernels — kept for A/B timing and counter
// studies (tools/, profiles/) — and the environment variables that select them (the windowed, v6 and row-mapped generations of
// rounds 2-4 left the tree in round 5: git tag r4-kernel-generations); nothing below an `#ifdef LZF_ANALYSIS` is in the product.
#pragma once
#include "lzf_device.h"

namespace lzf {
// perm (optional, everywhere below): launch index -> job index; capi.hip launches large batches longest job first
#ifdef LZF_ANALYSIS
__global__ void lzf_decompress_wave_kernel(const lzf_decompress_job* __restrict__ jobs,
                                           lzf_job_result* __restrict__ results, uint32_t n_jobs, const uint32_t* __restrict__ perm);
#endif
template <int RING, int S, int TOKCAP, bool STAGE>
__global__ __launch_bounds__(64) void lzf_decompress_batched_kernel(const lzf_decompress_job* __restrict__ jobs,
                                              lzf_job_result* __restrict__ results, uint32_t n_jobs, const uint32_t* __restrict__ perm);
// Tuning variants of the batched kernel: X(name, ring bytes, region bytes, token-list entries, chunk staged in LDS).
// LZF_DECOMPRESS_KERNEL=<name> selects one (A/B knob; every variant implements the same contract).
#ifdef LZF_ANALYSIS
#define LZF_DECOMPRESS_VARIANTS(X) \
    X(staged16, 4096, 16, 256, true)    \
    X(staged32, 4096, 32, 512, true)    \
    X(direct4w, 4096, 256, 2048, false)
#else
#define LZF_DECOMPRESS_VARIANTS(X) X(staged16, 4096, 16, 256, true)
#endif
#define LZF_EXT(NAME, R, S_, T, ST) extern template __global__ void lzf_decompress_batched_kernel<R, S_, T, ST>(const lzf_decompress_job*, lzf_job_result*, uint32_t, const uint32_t*);
LZF_DECOMPRESS_VARIANTS(LZF_EXT)
#undef LZF_EXT
// Producer / consumer pairs (lz4_decompress_paired.hip): X(name, ring bytes, region bytes, token-list entries).
// (The launch bounds are repeated on the template DECLARATIONS of this header since round 4: hipcc takes a template kernel's
//  attributes from its first declaration, and without them the instantiations were compiled for 1 024-thread workgroups.)
struct seg_job;
// done (optional): state array of the segmented pipeline; a job it finished (done[jid].done != 0) is skipped
template <int RING, int S, int TOKCAP>
__global__ __launch_bounds__(128) void lzf_decompress_paired_kernel(const lzf_decompress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs,
                                             const uint32_t* __restrict__ perm, const seg_job* __restrict__ done);
#ifdef LZF_ANALYSIS
#define LZF_PAIRED_VARIANTS(X) \
    X(paired16, 4096, 16, 256) \
    X(paired24, 4096, 24, 384) \
    X(paired48, 4096, 48, 640) \
    X(paired128, 4096, 128, 1280) \
    X(paired256, 4096, 256, 2048)
#else
#define LZF_PAIRED_VARIANTS(X) \
    X(paired24, 4096, 24, 384) \
    X(paired48, 4096, 48, 640)
#endif
#define LZF_EXTP(NAME, RG, S_, T) extern template __global__ void lzf_decompress_paired_kernel<RG, S_, T>(const lzf_decompress_job*, lzf_job_result*, uint32_t, const uint32_t*, const seg_job*);
LZF_PAIRED_VARIANTS(LZF_EXTP)
#undef LZF_EXTP
// ---------------------------------------------------------------------------------------------------------------------
// Segmented decompress (lz4_decompress_seg.hip): one block decoded by MANY wavefronts, as a pipeline of launches over the
// whole batch.  Used for batches that leave the chip mostly empty with one workgroup per block (per-block latency regime).
//   plan      per job: eligible?  (no prefix, no existing output, size window)  chunk / tile counts
//   parse     one wave per 16 KiB chunk of compressed bytes (chunks overlap by 2 KiB): speculative region-walk parse from
//             the chunk's first byte -> bit map "a token of this chunk's chain starts here" + the chain's exit
//   seam      one wave per job: chunk h's chain is the true one from the position where the true chain (exit of chunk
//             h - 1) steps on one of its marked tokens; almost always that is the exit itself (the chains met inside the
//             overlap), otherwise a short walk patches the bit map
//   tilesum   one wave per 2 KiB tile of compressed bytes: tokens and output bytes of the tile
//   scan      one wave per job: exclusive sums over the tiles, record space from the arena
//   records   one wave per tile: decode every token, absolute output positions, error checks, LITERALS -> out, and per
//             batch of 64 sequences the dependency level of every match (level k copies only from bytes that are final
//             once levels < k are done), its sub-batch and length class: one 16-byte record per sequence, final form
//   resolve   two waves per job, the only serial stage: ring of the recent output in LDS (filled with the literals
//             already in place by the stager wave, flushed by it with aligned 16-byte stores), per batch one LDS round
//             per dependency level by the resolver wave (a hand-scheduled loop)
// A job that is not eligible, or in which any stage meets something it does not handle (every DecodeError, capacity,
// arena exhausted), is left to the pair kernel, which runs last and skips the jobs the pipeline finished.
// ---------------------------------------------------------------------------------------------------------------------
struct seg_job {
    uint32_t eligible, failed, done, nch;
    uint32_t ntile, ntok, outb, pad;
    uint64_t rec_off;            // first record of the job in the arena
    uint64_t pad2;
};
struct seg_ctx {
    const lzf_decompress_job* jobs;
    lzf_job_result* results;
    seg_job* st;
    uint32_t* bits;              // [n_jobs][maxch][512]   token bit maps of the chunks' chains
    uint32_t* xexit;             // [n_jobs][maxch]        first token at or beyond the chunk's end, on the chunk's chain
    uint32_t* vfrom;             // [n_jobs][maxch]        position from which the chunk's bits are the true tokens (~0: none)
    uint32_t* tile_tok;          // [n_jobs][maxtile]      tokens per tile, then (scan) first token of the tile
    uint32_t* tile_out;          // [n_jobs][maxtile]      output bytes per tile, then (scan) first output byte of the tile
    u32x4* recs;                 // arena of records
    unsigned long long* rec_top; // bump pointer into the arena
    uint64_t rec_cap;
    uint32_t n_jobs, maxch, maxtile, max_in, min_in;
    uint32_t ring_bytes;         // LDS ring of the resolve stage (32 / 64 / 128 KiB): the records stage classes the sequences for it
    uint32_t* order;             // [n_jobs]  resolve stage: workgroup -> job (null: identity); see lzf_seg_order_kernel
    uint32_t n_cu;               // compute units of the device (the order deals the jobs out in rows of this many)
    uint32_t* by_len;            // [n_jobs]  chunk / tile stages: grid row -> job, longest input first (null: identity)
    uint32_t rec_by_len;         // the records stage follows by_len too (64 jobs and more; measured: below that its own order is quicker)
    uint32_t dbg_force;          // analysis library only (LZF_SEG_FORCE): 1 = stagers of odd jobs give up, 2 = resolvers of odd jobs give up
    // A call's jobs may go through the last two stages in GROUPS (capi.hip: the next group's records stage runs under the resolve
    // stage of the one before): a launch then serves ranks [g_off, g_off + g_n) of by_len[] (which is the jobs by sequences then).
    // One group: g_off = 0, g_n = n_jobs, grouped = 0.
    uint32_t g_off, g_n, grouped;
    uint32_t res_prio;           // the resolve stage's wavefronts raise their issue priority (grouped calls: they share their SIMDs with the next group's records stage)
    uint32_t fed;                // the plan / parse / seam stages serve the bitmap-fed kernel: prefix and existing output do not make a job ineligible
};
// ---------------------------------------------------------------------------------------------------------------------
// Bitmap-fed decompress (lz4_decompress_fed.hip): batches beyond what the chip holds at once.  plan + parse + seam of the
// segmented pipeline run over the whole batch (seg_ctx::fed: a job's prefix / existing output do not matter to the parse),
// then one wavefront per block lists its tokens from the bit map — verifying the chain link by link — and copies.  A job the
// kernel does not finish cleanly stays !done for the pair kernel launched behind it.
// ---------------------------------------------------------------------------------------------------------------------
// decoder state a job's pieces hand on: flag = pieces finished so far (piece p starts when it reads p), kFedEnded once the job needs no more
struct fed_state { uint32_t flag, cstart, expect, o; };
constexpr uint32_t kFedEnded = 0x80000000u;
constexpr uint32_t kFedTicketStride = 32u;      // (a counter per 128-byte line)
struct fed_args {
    const lzf_decompress_job* jobs;
    lzf_job_result* results;
    seg_job* st;
    const uint32_t* bits;        // seg_ctx::bits
    const uint32_t* vfrom;       // seg_ctx::vfrom
    const uint32_t* perm;        // launch order: rank -> job (optional)
    fed_state* state;            // [n_jobs]
    uint32_t* ticket;            // the launch's ticket counters, one per XCD, kFedTicketStride words apart
    uint32_t xcc_mask;           // the XCDs the kernel's wavefronts run on (census): bit i = HW_REG_XCC_ID i
    uint32_t n_jobs, maxch;
    uint32_t pieces;             // every job goes through the kernel in this many pieces (1: whole)
    uint32_t* census;            // not null: the launch only counts how many of its workgroups the device holds at once ([0] arrivals, [1] the answer, [2] mask of their XCDs)
};
__global__ void lzf_fed_reset_kernel(fed_args a);
// X(name, ring bytes, bit-map words per round, token-list entries)
#define LZF_FED_VARIANTS(X) X(fed32, 4096, 32, 352)
template <int RING, int W, int TOKCAP>
__global__ __launch_bounds__(64) void lzf_decompress_fed_kernel(fed_args a);
#define LZF_EXTF(NAME, RG, W_, T) extern template __global__ void lzf_decompress_fed_kernel<RG, W_, T>(fed_args);
LZF_FED_VARIANTS(LZF_EXTF)
#undef LZF_EXTF
// grid row / workgroup index of a group's launch -> job
__device__ __forceinline__ uint32_t seg_job_of(const seg_ctx& c, uint32_t i) { return c.by_len ? c.by_len[i + c.g_off] : i + c.g_off; }
constexpr uint32_t kSegRegion = 256, kSegChunk = 64u * kSegRegion, kSegOverlap = 2048, kSegStride = kSegChunk - kSegOverlap;
constexpr uint32_t kSegChunkWords = kSegChunk / 32u, kSegTile = 2048;
__global__ void lzf_seg_plan_kernel(seg_ctx c);
// batches of more than one block per CU: which workgroup of the resolve stage takes which job — the jobs ranked by their number of
// sequences and dealt out in rows of n_cu, every other row reversed, so that the blocks that share a CU (workgroups k, k + n_cu,
// k + 2 n_cu ... land on the same CU) are a slow one with fast ones: the launch ends with its slowest CU
__global__ void lzf_seg_order_kernel(seg_ctx c);
__global__ void lzf_seg_pause_kernel(uint32_t ticks);
__global__ void lzf_seg_rank_kernel(seg_ctx c, uint32_t* __restrict__ by_tok, uint4 group_sizes);
// the same batches: grid row -> job for the chunk / tile stages, longest input first (a launch ends with its last rows)
__global__ void lzf_seg_by_len_kernel(seg_ctx c);
__global__ void lzf_seg_parse_kernel(seg_ctx c);
__global__ void lzf_seg_seam_kernel(seg_ctx c);
__global__ void lzf_seg_tilesum_kernel(seg_ctx c);
__global__ void lzf_seg_scan_kernel(seg_ctx c);
__global__ void lzf_seg_records_kernel(seg_ctx c);
template <int R>
__global__ __launch_bounds__(128) void lzf_seg_resolve_pair_kernel(seg_ctx c);
extern template __global__ void lzf_seg_resolve_pair_kernel<32768>(seg_ctx);
extern template __global__ void lzf_seg_resolve_pair_kernel<65536>(seg_ctx);
extern template __global__ void lzf_seg_resolve_pair_kernel<131072>(seg_ctx);
template <int KIND>
__global__ __launch_bounds__(64) void lzf_compress_wave_kernel(const lzf_compress_job* __restrict__ jobs,
                                         lzf_job_result* __restrict__ results, uint32_t n_jobs, uint32_t skip_compact,
                                         const uint32_t* __restrict__ perm);
extern template __global__ void lzf_compress_wave_kernel<LZF_TABLE_U32>(const lzf_compress_job*, lzf_job_result*, uint32_t, uint32_t, const uint32_t*);
extern template __global__ void lzf_compress_wave_kernel<LZF_TABLE_U16>(const lzf_compress_job*, lzf_job_result*, uint32_t, uint32_t, const uint32_t*);
// LDS of one workgroup of lzf_compress_compact_kernel: 2048 words of slots + 128 of epoch parities + one scratch word per lane (the dispatch derives its residency from this)
constexpr uint32_t kCompactLdsBytes = (4096u / 2u + 4096u / 32u + 64u) * 4u;
template <bool DRY>
__global__ __launch_bounds__(64) void lzf_compress_compact_kernel(const lzf_compress_job* __restrict__ jobs,
                                            lzf_job_result* __restrict__ results, uint32_t n_jobs, const uint32_t* __restrict__ perm,
                                            uint32_t alone);
extern template __global__ void lzf_compress_compact_kernel<false>(const lzf_compress_job*, lzf_job_result*, uint32_t, const uint32_t*, uint32_t);
extern template __global__ void lzf_compress_compact_kernel<true>(const lzf_compress_job*, lzf_job_result*, uint32_t, const uint32_t*, uint32_t);
// the latency class (lz4_compress_team.hip / .inc; round 5): one block per CU, searcher / emitter / feeder wavefronts, input ring and table in LDS
__global__ void lzf_compress_team_kernel(const lzf_compress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs,
                                         const uint32_t* __restrict__ perm, uint32_t alone);
__global__ void lzf_compress_team_carry_kernel(const lzf_compress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs, const uint32_t* __restrict__ perm);
// job ordering (aux_kernels.hip): cost probes of the compress jobs and the launch order derived from them
__global__ void lzf_cost_probe_jobs_kernel(const lzf_compress_job* __restrict__ jobs, lzf_compress_job* __restrict__ probes, uint32_t n,
                                           uint32_t piece, uint32_t parts);
__global__ void lzf_order_by_input_len_kernel(const lzf_decompress_job* __restrict__ jobs, uint32_t* __restrict__ perm, uint32_t n);
__global__ void lzf_decompress_cost_kernel(const lzf_decompress_job* __restrict__ jobs, uint32_t n, uint32_t* __restrict__ est, uint32_t len_shift);
__global__ void lzf_order_by_estimate_kernel(const uint32_t* __restrict__ est, uint32_t* __restrict__ perm, uint32_t n);
__global__ void lzf_order_by_cost_kernel(const lzf_compress_job* __restrict__ jobs, const lzf_job_result* __restrict__ probe_results,
                                         uint32_t* __restrict__ perm, uint32_t n, uint32_t piece, uint32_t parts);
__global__ void lzf_xxh32_kernel(const uint8_t* const* __restrict__ ptrs, const uint64_t* __restrict__ lens,
                                 uint32_t* __restrict__ out, uint32_t n);
__global__ void lzf_xxh32_wave_kernel(const uint8_t* const* __restrict__ ptrs, const uint64_t* __restrict__ lens,
                                      uint32_t* __restrict__ out, uint32_t n);
__global__ void lzf_copy_ranges_kernel(const uint8_t* const* __restrict__ src, uint8_t* const* __restrict__ dst,
                                       const uint64_t* __restrict__ len, uint32_t n);
__global__ void lzf_seed_table_kernel(lzf_u32_table* __restrict__ t, const uint8_t* __restrict__ dict, uint64_t dict_len);
__global__ void lzf_table_offset_kernel(void* table, uint32_t kind, uint64_t add);
__global__ void lzf_table_offset_batch_kernel(void* const* __restrict__ tables, const uint64_t* __restrict__ adds, uint32_t n, uint32_t kind);
__global__ void lzf_chain_decompress_step_kernel(const lzf_chain_step* __restrict__ steps, lzf_chain_state* __restrict__ state, uint32_t n,
                                                 lzf_decompress_job* __restrict__ jobs, const lzf_job_result* __restrict__ results);
}  // namespace lzf
