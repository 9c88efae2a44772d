// lz4_decompress_fed.hip — raw::decompress_raw (src/raw/decompress.rs:58-138) for batches beyond what the chip holds at once,
// with the PARSE taken out of the block's own wavefronts: the token positions of every block of the batch come from the hop
// parse of the segmented pipeline (lzf_seg_parse_kernel + lzf_seg_seam_kernel: one bit per compressed byte, 3.8
// wave-instructions per sequence against the 13.0 of the in-kernel region parse of lz4_decompress_paired.hip), and the kernel
// here only FEEDS its copy stage from that map (lz4_decompress_feed_phase.inc: bit map -> token list, lengths pre-decoded, the
// chain verified link by link) and runs the unchanged COPY stage (lz4_decompress_batch_phase.inc).
//
//   lzf_decompress_fed_kernel<RING, W, TOKCAP>        one wavefront per block: feed a round of 32 * W compressed bytes, copy it
//   lzf_decompress_fed_pair_kernel<RING, W, TOKCAP>   two per block: wave 0 feeds round k + 1 while wave 1 copies round k
//
// Contract with the dispatch (capi.hip): a job is taken only when the plan stage found it eligible (sizes inside the bit map's
// window) and the seam stage did not fail on it; a job is FINISHED here (results written, seg_job::done set) only when it decodes
// cleanly over a verified chain.  Everything else — every DecodeError, a capacity problem, a chain that does not verify — is
// left untouched for the pair kernel launched behind this one, which decodes the job from its first byte and reports the
// reference's status.  What this kernel wrote into `out` before it gave up is a prefix of what that kernel writes again.
#include "lzf_device.h"
#include "kernels.h"
#include "lzf_copy_helpers.h"

namespace lzf {

// the XCD this wavefront runs on (0..7 on MI355X)
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u;
}

// ---------------------------------------------------------------------------------------------------------------------
// The launch is a fixed set of SLOTS — as many wavefronts as the device holds at once (capi.hip counts them) — that share the
// jobs out in PIECES instead of one workgroup per job.  With one workgroup per job a launch of 2.2 jobs per slot takes three
// rounds of the longest jobs' time, and inside a round every slot waits for the longest job (the timeline of round 6: 86 ms for
// 66 ms of work; a block costs 42 .. 86 M cycles and nothing cheap predicts which).  Here every job is cut into `pieces`
// stretches of equal compressed length, and the slots draw TICKETS from one counter: ticket t is piece t / n of the job of rank
// t % n in launch order — lap after lap over all jobs, so all of them advance together and end together, and a slot that gets
// cheap pieces simply draws more tickets.  A piece starts from the decoder state the piece before it parked (next round, chain
// carry, output position; the ring is re-filled from `out`) and waits for it if it has to: it was drawn n tickets earlier, and
// a launch has more jobs than slots, so as a rule it is long done.
// Hand-overs stay inside one XCD: the wavefronts read which XCD they run on (HW_REG_XCC_ID) and every XCD has its own ticket
// counter over its own share of the jobs (rank % number of XCDs).  What a piece wrote is then in the L2 its successor reads
// through — a wait for the stores and an L1 invalidate are the whole hand-over.  Across XCDs it would take a write-back of the
// writer's whole L2 (buffer_wbl2: measured ~150 us of the wave per hand-over, 98 -> 115 ms per call at 64 pieces per job).
// ---------------------------------------------------------------------------------------------------------------------
template <int RING, int W, int TOKCAP>
__global__ __launch_bounds__(64) void lzf_decompress_fed_kernel(fed_args a) {
    constexpr bool STAGE = true;
    constexpr uint32_t kMask = RING - 1;
    constexpr uint32_t kSpanMax = RING / 3;            // output bytes one batch may produce
    constexpr uint32_t kNearHist = RING - kSpanMax;    // history before the batch that stays intact in the ring
    constexpr uint32_t kRound = 32u * (uint32_t)W;     // compressed bytes whose tokens one round lists
    constexpr uint32_t kCB = kRound + 128u;            // staged bytes: the round + room for the bodies of its last tokens
    static_assert(W >= 1 && W <= 64 && (kSegTile % kRound) == 0, "a round is one bit-map word per lane and subdivides a tile");
    static_assert(TOKCAP >= (int)(kRound / 3u + 1u), "a round's tokens (at least three bytes each) fit the list");
    static_assert(kCB % 16 == 0, "the round is staged in 16-byte pieces");
    __shared__ __attribute__((aligned(16))) uint8_t ring[RING];
    __shared__ __attribute__((aligned(16))) uint8_t cbuf[kCB];
    __shared__ __attribute__((aligned(16))) uint32_t toks[TOKCAP];

    const uint32_t lane = threadIdx.x;
    if (a.census) {
        // How many workgroups of THIS kernel (its LDS, registers) does the device hold at once?  Every workgroup counts itself in,
        // stays for 300 us and reads the count when it leaves: the first residents read the residency, later ones more; the minimum
        // is the answer.  (The schedule needs the true number: a slot that starts late finishes late, and the slot that waits for
        // its hand-over with it.)
        if (lane == 0u) {
            __hip_atomic_fetch_or(&a.census[2], 1u << (xcc_id() & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... and on which XCDs they run
            __hip_atomic_fetch_add(&a.census[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = wall_clock64();
            while (wall_clock64() - t0 < 30000ull) __builtin_amdgcn_s_sleep(16);
            const uint32_t seen = __hip_atomic_load(&a.census[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_min(&a.census[1], seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const uint32_t n = a.n_jobs, pieces = a.pieces;
    // this wavefront's XCD -> its group of jobs (ranks g, g + ng, g + 2 ng, ...) and that group's ticket counter
    // (whole jobs need no hand-over: one group, one counter)
    const uint32_t xcc = xcc_id() & 31u;
    if (pieces > 1u && !((a.xcc_mask >> xcc) & 1u)) return;           // (an XCD the census did not see: no jobs were given to it)
    const uint32_t ng = pieces > 1u ? (uint32_t)__popc(a.xcc_mask) : 1u, g = pieces > 1u ? (uint32_t)__popc(a.xcc_mask & ((1u << xcc) - 1u)) : 0u;
    const uint32_t n_g = n > g ? (n - g + ng - 1u) / ng : 0u;
    const uint32_t n_tickets = n_g * pieces;
    for (;;) {
    uint32_t tk = 0;
    if (lane == 0u) tk = __hip_atomic_fetch_add(a.ticket + g * kFedTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tk = __builtin_amdgcn_readfirstlane(tk);
    if (tk >= n_tickets) break;
    const uint32_t piece = tk / n_g, k = (tk - piece * n_g) * ng + g;
    const uint32_t jid = a.perm ? a.perm[k] : k;
    const seg_job sj = a.st[jid];
    if (!sj.eligible || sj.failed || sj.done) continue;
    LZF_GLOBAL fed_state* const fs = (LZF_GLOBAL fed_state*)a.state + jid;
    const lzf_decompress_job job = a.jobs[jid];
    const long long t_start = clock64();
#ifdef LZF_DBG_TIMELINE
    const unsigned long long t_wall0 = wall_clock64();
#endif

    int status = LZF_OK;
    bool bailed = false, parked = false;
    uint32_t o = 0;
#ifdef LZF_DBG_PHASE_SEL
    long long ph_acc_out = 0;
#endif
    if (job.input_len >= kMaxPosB || job.out_existing_len >= kMaxPosB || job.prefix_len >= kMaxPosB || job.out_existing_len > job.out_cap) continue;   // (LZF_CONTRACT: the pair kernel says so)
    {
        cgu8* __restrict__ in = as_global(job.input);
        cgu8* __restrict__ prefix = as_global(job.prefix);
        gu8* out = as_global(job.out);
        const uint32_t len = (uint32_t)job.input_len;
        const uint32_t plen = (uint32_t)job.prefix_len;
        const uint32_t cap = job.out_cap > kMaxPosB ? kMaxPosB : (uint32_t)job.out_cap;
        const uint64_t limit = job.output_limit;
        const uint32_t rb = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u);   // ring bias
        const uint32_t ring_a = lds_addr(ring), cbuf_a = lds_addr(cbuf);
        const LZF_GLOBAL uint32_t* const fed_bits = (const LZF_GLOBAL uint32_t*)a.bits + (size_t)jid * a.maxch * kSegChunkWords;
        const LZF_GLOBAL uint32_t* const fed_vf = (const LZF_GLOBAL uint32_t*)a.vfrom + (size_t)jid * a.maxch;
#define RIDX(x) (((x) + rb) & kMask)

        // ring <- out[a, b)   (b - a <= RING; caller made out[a,b) visible)
        auto ring_fill = [&](uint32_t a_, uint32_t b) {
            uint32_t nh = (16u - ((a_ + rb) & 15u)) & 15u; if (nh > b - a_) nh = b - a_;
            if (lane < nh) ring[RIDX(a_ + lane)] = out[a_ + lane];
            a_ += nh;
            const uint32_t nchunks = (b - a_) >> 4;
            for (uint32_t c = lane; c < nchunks; c += kWave)
                *reinterpret_cast<u32x4*>(&ring[RIDX(a_ + 16u * c)]) = *reinterpret_cast<const LZF_GLOBAL u32x4*>(out + a_ + 16u * c);
            a_ += nchunks << 4;
            if (lane < b - a_) ring[RIDX(a_ + lane)] = out[a_ + lane];
        };
        // out[a, b) <- ring
        auto ring_flush = [&](uint32_t a_, uint32_t b) {
            uint32_t nh = (16u - ((a_ + rb) & 15u)) & 15u; if (nh > b - a_) nh = b - a_;
            if (lane < nh) out[a_ + lane] = ring[RIDX(a_ + lane)];
            a_ += nh;
            const uint32_t nchunks = (b - a_) >> 4;
            for (uint32_t c = lane; c < nchunks; c += kWave)
                *reinterpret_cast<LZF_GLOBAL u32x4*>(out + a_ + 16u * c) = *reinterpret_cast<const u32x4*>(&ring[RIDX(a_ + 16u * c)]);
            a_ += nchunks << 4;
            if (lane < b - a_) out[a_ + lane] = ring[RIDX(a_ + lane)];
        };

        uint32_t expect = 0;                 // where the next token of the chain starts (len: the chain has ended)
        uint32_t cstart = 0;
        o = (uint32_t)job.out_existing_len;
        // this piece: rounds [piece, piece + 1) * per of the job's input
        const uint32_t rounds = (len + kRound - 1u) / kRound, per = (rounds + pieces - 1u) / pieces;
        const uint32_t piece_end = (piece + 1u) * per >= rounds ? len : (piece + 1u) * per * kRound;
        if (piece > 0u) {
            // the piece before this one was drawn n tickets ago: as a rule it is done; else wait for it (bounded), then take its state over
            uint32_t f = 0;
            for (uint32_t spin = 0; spin < (1u << 22); ++spin) {
                f = __hip_atomic_load(&fs->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (f >= piece) break;
                __builtin_amdgcn_s_sleep(32);
            }
            f = __builtin_amdgcn_readfirstlane(f);
            if (f != piece) continue;        // the job ended in an earlier piece (kFedEnded), or that piece never came: the pair kernel looks at what is left
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // (this compute unit's L1 may hold lines of `out` from an earlier piece of the job)
            cstart = fs->cstart; expect = fs->expect; o = fs->o;
        }
        uint32_t safe = o;   // out[0, safe) is visible to this wave's global loads
        if (o > 0) ring_fill(o > (uint32_t)RING ? o - RING : 0u, o);   // Vec content on entry (or what the other slot wrote) = history
#ifdef LZF_DBG_PHASE_SEL   // analysis: cycles the wave spends in section LZF_DBG_PHASE_SEL (section i ends at PHASE(i); 0-5: the copy stage's, lz4_decompress_paired.hip;
                           // 6 stage + bit map, 7 bit map -> list, 8 lengths + chain check) -> results[].reserved
        long long ph_t = clock64(), ph_acc = 0;
#define PHASE(i) do { const long long tn__ = clock64(); if ((i) == LZF_DBG_PHASE_SEL) ph_acc += tn__ - ph_t; ph_t = tn__; } while (0)
#else
#define PHASE(i) do { } while (0)
#endif
        while (cstart < len && expect < len && status == LZF_OK) {
            if (cstart >= piece_end) { parked = true; break; }      // the next piece's
            __syncthreads();                 // (one wave: orders the re-use of cbuf / toks between rounds)
#include "lz4_decompress_feed_phase.inc"
            if (bail) { bailed = true; break; }
            if (Tc) {
#define LZF_TOKEN_AT(i) (toks[(i)] & 0xFFFFu)
#define LZF_TOKEN_WORD(i) toks[(i)]
#ifdef LZF_FED_FAR_LATE
#define LZF_FAR_LATE
#endif
#include "lz4_decompress_batch_phase.inc"
#undef LZF_FAR_LATE
#undef LZF_TOKEN_WORD
#undef LZF_TOKEN_AT
            }
            // the next round the chain has a token in
            const uint32_t nx = expect & ~(kRound - 1u);
            cstart = nx > cstart ? nx : cstart + kRound;
        }
        if (!parked && status == LZF_OK && expect < len) bailed = true;      // the map ends before the chain does
#ifdef LZF_DBG_PHASE_SEL
        ph_acc_out = ph_acc;
#endif
#undef PHASE
#undef RIDX
        if (parked && status == LZF_OK && !bailed) {
            // hand the job on: what this wave wrote must be visible to another compute unit before the flag is
            if (lane == 0u) { fs->cstart = cstart; fs->expect = expect; fs->o = o; }
            // (no release fence: the next piece runs on this XCD and reads through the same L2 — the stores only have to have arrived there)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0u) __hip_atomic_store(&fs->flag, piece + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
    }
    if (pieces > 1u && lane == 0u) __hip_atomic_store(&fs->flag, kFedEnded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // finished or given up: later tickets of the job pass
    if (bailed || status != LZF_OK) continue;        // not ours: the pair kernel decodes the job again and reports the status
    if (lane == 0) {
        a.results[jid].out_len = o;
        a.results[jid].status = LZF_OK;
#ifdef LZF_DBG_PHASE_SEL
        a.results[jid].reserved = (uint32_t)(ph_acc_out >> 10);
#elif defined(LZF_DBG_TIMELINE)   // analysis: when the job ran, on the 100 MHz wall clock every wave reads alike (units of 2.56 us): start << 16 | end
        a.results[jid].reserved = (uint32_t)(((t_wall0 >> 8) & 0xFFFFull) << 16) | (uint32_t)((wall_clock64() >> 8) & 0xFFFFull);
#else
        a.results[jid].reserved = (uint32_t)((clock64() - t_start) >> 10);   // diagnostic: shader kilo-cycles spent on this job
#endif
        a.st[jid].done = 1u;
    }
    }
}

// Before the launch: every job's hand-over flag and the ticket counter cleared.
__global__ __launch_bounds__(256) void lzf_fed_reset_kernel(fed_args a) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < 32u) a.ticket[k * kFedTicketStride] = 0u;
    if (k < a.n_jobs) a.state[k].flag = 0u;
}

#define LZF_INSTF(NAME, RG, W_, T) template __global__ void lzf_decompress_fed_kernel<RG, W_, T>(fed_args);
LZF_FED_VARIANTS(LZF_INSTF)
#undef LZF_INSTF

}  // namespace lzf
