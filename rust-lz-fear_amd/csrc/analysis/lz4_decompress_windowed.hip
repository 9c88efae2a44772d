// lz4_decompress_windowed.hip — batched raw::decompress_raw for gfx950, third generation.
//
// Same contract and the same COPY stage as lz4_decompress_batched.hip (src/raw/decompress.rs:58-138,
// one wavefront per block); what changes is the PARSE.  Counters showed the second generation to be
// bound by instruction issue (≈2.7 wave-instructions/ns/CU, the ceiling tools/issue_mix_microbench.hip
// measures for a SALU/VALU mix), with the speculative parse costing ≈15 of its ≈33 instructions per
// sequence: small regions (so that chunk, tables and token list fit in LDS) need ≈20 fixed-point passes.
//
// Here a chunk is 64 regions of R = 512 bytes.  With regions that long a walk started from a guessed
// position is almost always back on the true token chain before it leaves the warm-up stretch, so the
// fixed point  start[i] <- max(exit[0..i-1])  settles in 2-3 passes and a token costs about one hop per
// walk — 64 lanes hop at once, so ≈0.1 wave-iterations per token.  What makes long regions affordable:
//   * each lane reads its token bytes through a private 64-byte WINDOW in LDS, refilled from HBM/L2 with
//     four 16-byte loads per three hops (a plain token advances at most 19 bytes) — the scattered
//     4-byte and 1-byte loads of a per-hop global read had saturated the L1/TA path (TCP counters in
//     profiles/) long before anything else;
//   * the hop itself is straight-line predicated VALU code (no exec-mask control flow, 0 SALU);
//   * the token list (u16 chunk offsets) goes to a per-block scratch area in HBM (it is written once
//     and read back once, coalesced, and stays in L2), so LDS holds only the output ring and the
//     windows: 8 KB per wave.
// decompress.rs:61-71 (token, LSIC lengths, literals, offset) is what a hop steps over; tokens the plain
// view cannot express (0xFF length bytes, literal runs longer than the window, the last 24 bytes of the
// input) are taken one at a time by the general routine.
#include "../lzf_device.h"
#include "../kernels.h"
#include "../lzf_copy_helpers.h"

namespace lzf {

namespace {

constexpr uint32_t kWarm = 128;      // bytes a first-guess walk runs before its region

// One predicated hop on LDS window coordinates (all operands per lane):
//   p      LDS address of the current token (window base + position - window start)
//   lim    the lane hops while p < lim (0: idle)
//   fend   a hop is taken only if the byte after the offset field lies below fend (inside the window, inside the input margin)
//   pclamp last address a 4-byte read may start at
// q is forced to ~0 when the token needs more than this view (a 0xFF length byte).
#define LZF_WHOP_HEAD \
    "v_min_u32 %[pa], %[pclamp], %[p]\n\t" \
    "ds_read_b32 %[w], %[pa]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_bfe_u32 %[t], %[w], 4, 4\n\t"                 /* literal-length nibble */ \
    "v_bfe_u32 %[q], %[w], 8, 8\n\t"                 /* first extension byte */ \
    "v_cmp_eq_u32 vcc, 15, %[t]\n\t" \
    "v_add_u32 %[q], 1, %[q]\n\t" \
    "v_cndmask_b32 %[q], 0, %[q], vcc\n\t" \
    "v_add3_u32 %[q], %[pa], %[t], %[q]\n\t" \
    "v_add_u32 %[q], 3, %[q]\n\t"                    /* first byte after the offset */ \
    "v_and_b32 %[t], 0xfff0, %[w]\n\t" \
    "v_cmp_eq_u32 vcc, 0xfff0, %[t]\n\t"             /* nibble 15 and extension 0xFF */ \
    "v_cndmask_b32_e64 %[q], %[q], -1, vcc\n\t" \
    "v_and_b32 %[t], 15, %[w]\n\t"                   /* match-length nibble */ \
    "v_cmp_gt_u32 vcc, %[fend], %[q]\n\t" \
    "v_cndmask_b32 %[t], 0, %[t], vcc\n\t" \
    "v_cmp_eq_u32 vcc, 15, %[t]\n\t"                 /* needs the first match-length extension byte */ \
    "v_cndmask_b32 %[m], %[pa], %[q], vcc\n\t" \
    "ds_read_u8 %[m], %[m]\n\t" \
    "v_addc_co_u32_e64 %[q], %[sx], 0, %[q], vcc\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_cndmask_b32 %[m], 0, %[m], vcc\n\t" \
    "v_cmp_eq_u32 vcc, 0xff, %[m]\n\t" \
    "v_cndmask_b32_e64 %[q], %[q], -1, vcc\n\t" \
    "v_cmp_lt_u32 vcc, %[p], %[lim]\n\t" \
    "v_cndmask_b32 %[t], -1, %[q], vcc\n\t" \
    "v_cmp_gt_u32 vcc, %[fend], %[t]\n\t"            /* vcc = the lane takes this hop */
#define LZF_WHOP_TAIL \
    "v_addc_co_u32_e64 %[n], %[sx], 0, %[n], vcc\n\t" \
    "v_cndmask_b32 %[p], %[p], %[q], vcc\n\t"
// recording form: token position (chunk offset = pa + delta) -> gtoks[k++]; lanes that do not hop store to their dump slot
#define LZF_WHOP_RECORD \
    "v_cndmask_b32 %[kk], %[dump], %[k], vcc\n\t" \
    "v_addc_co_u32_e64 %[k], %[sx], 0, %[k], vcc\n\t" \
    "v_lshlrev_b32 %[kk], 1, %[kk]\n\t" \
    "v_add_u32 %[t], %[pa], %[delta]\n\t" \
    "global_store_short %[kk], %[t], %[gt]\n\t"

__device__ __forceinline__ void whop3(uint32_t& p, uint32_t lim, uint32_t& n, uint32_t fend, uint32_t pclamp) {
    uint32_t pa, w, q, t, m; uint64_t sx;
    asm volatile(LZF_WHOP_HEAD LZF_WHOP_TAIL LZF_WHOP_HEAD LZF_WHOP_TAIL LZF_WHOP_HEAD LZF_WHOP_TAIL
                 : [p] "+v"(p), [n] "+v"(n), [pa] "=&v"(pa), [w] "=&v"(w), [q] "=&v"(q), [t] "=&v"(t), [m] "=&v"(m), [sx] "=&s"(sx)
                 : [lim] "v"(lim), [fend] "v"(fend), [pclamp] "v"(pclamp)
                 : "vcc", "memory");
}
__device__ __forceinline__ void whop3_record(uint32_t& p, uint32_t lim, uint32_t& n, uint32_t fend, uint32_t pclamp,
                                             uint32_t& k, uint32_t dump, uint32_t delta, LZF_GLOBAL uint16_t* gt) {
    uint32_t pa, w, q, t, m, kk; uint64_t sx;
    asm volatile(LZF_WHOP_HEAD LZF_WHOP_RECORD LZF_WHOP_TAIL LZF_WHOP_HEAD LZF_WHOP_RECORD LZF_WHOP_TAIL LZF_WHOP_HEAD LZF_WHOP_RECORD LZF_WHOP_TAIL
                 : [p] "+v"(p), [n] "+v"(n), [k] "+v"(k), [pa] "=&v"(pa), [w] "=&v"(w), [q] "=&v"(q), [t] "=&v"(t), [m] "=&v"(m),
                   [kk] "=&v"(kk), [sx] "=&s"(sx)
                 : [lim] "v"(lim), [fend] "v"(fend), [pclamp] "v"(pclamp), [dump] "v"(dump), [delta] "v"(delta), [gt] "s"(gt)
                 : "vcc", "memory");
}

}  // namespace

template <int RING, int R, int WIN>
__global__ __launch_bounds__(64) void lzf_decompress_windowed_kernel(
    const lzf_decompress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs,
    uint16_t* __restrict__ scratch, uint32_t scratch_stride) {
    constexpr uint32_t kWin = WIN;                     // bytes of a lane's window (64 or 128)
    constexpr uint32_t kMask = RING - 1;
    constexpr uint32_t kSpanMax = RING / 3;            // output bytes one batch may produce
    constexpr uint32_t kNearHist = RING - kSpanMax;    // history before the batch that stays intact in the ring
    constexpr uint32_t kChunk = 64u * R;               // compressed bytes whose tokens one parse covers
    constexpr uint32_t kCap = kChunk / 3u + 1u;        // most tokens a chunk can hold (a token is at least 3 bytes)
    static_assert(kChunk <= 65536, "token positions are stored as u16 offsets into the chunk");
    constexpr bool STAGE = false;                      // the COPY stage reads tokens and literals from HBM/L2
    constexpr uint32_t kCB = 0; const uint32_t cbuf_a = 0;
    __shared__ __attribute__((aligned(16))) uint8_t ring[RING];
    __shared__ __attribute__((aligned(16))) uint8_t win[64u * kWin + 16u];

    const uint32_t jid = blockIdx.x;
    if (jid >= n_jobs) return;
    const uint32_t lane = threadIdx.x;
    const lzf_decompress_job job = jobs[jid];
    const long long t_start = clock64();
    LZF_GLOBAL uint16_t* gtoks = (LZF_GLOBAL uint16_t*)(scratch + (size_t)jid * scratch_stride);   // kCap + 64 entries

    int status = LZF_OK;
    uint32_t o = 0;
#ifdef LZF_DBG_COUNT
    uint32_t dbg_rounds = 0, dbg_slow = 0, dbg_walks = 0;      // analysis builds only (tools/dbg_counts.py)
#define LZF_COUNT(x) (++(x))
#else
#define LZF_COUNT(x) ((void)0)
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
        const uint32_t ring_a = lds_addr(ring), win_a = lds_addr(win);
#define RIDX(x) (((x) + rb) & kMask)
#define PHASE(i) do { } while (0)

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

        // plain hops stay clear of the input's end; short inputs are parsed by the general routine alone
        const uint32_t fast_end = len >= 2u * kWin ? len - 24u : 0u;
        const uint32_t wl = win_a + lane * kWin;                      // this lane's window
        auto rdb = [&](uint32_t q) -> uint32_t { return (uint32_t)in[q]; };
        // 4 input bytes at q (missing bytes past the end read as 0)
        auto rd4 = [&](uint32_t q) -> uint32_t {
            if (q + 4u <= len) return ld4(in + q);
            uint32_t v = 0;
            for (uint32_t i = 0; i < 4u && q + i < len; ++i) v |= rdb(q + i) << (8u * i);
            return v;
        };
        // One token at p (p < len): position of the next token; false on UnexpectedEnd.
        // decompress.rs:61-71 without the copies.
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

        uint32_t cstart = 0;                 // a true token position (or len)
        while (cstart < len && status == LZF_OK) {
            // Walk the token chain from p up to (not including) the first token at or beyond `end`.
            // Tokens are counted in n and, when RECORD, their chunk offsets go to gtoks[k++].
            auto walk = [&](uint32_t p, const uint32_t end, uint32_t& n, uint32_t& k, bool& err, bool go, auto RECORD) -> uint32_t {
                const uint32_t stop = end < fast_end ? end : fast_end;
                LZF_COUNT(dbg_walks);
                for (;;) {
                    const bool live = go && p < stop;
                    bool stuck = false;
                    if (__any(live)) {
                        LZF_COUNT(dbg_rounds);
                        // refill: the window starts at the lane's position (kept inside the input), then three hops
                        const uint32_t p0 = p;
                        const uint32_t wb = p < len - kWin ? p : len - kWin;
                        {
                            // Transposed fetch: lane l moves piece (l & 3) of the windows of lanes (l >> 2) + 16 j, so one
                            // load instruction touches 16 windows' cache lines instead of 64 (the L1 looks lines up one a cycle).
                            const unsigned long long lm = __ballot(live);
                            constexpr uint32_t kPieces = kWin / 16u;              // 16-byte pieces per window = lanes per window
                            constexpr uint32_t kPerInstr = 64u / kPieces;         // windows one instruction serves
                            const uint32_t piece = (lane % kPieces) * 16u;
                            u32x4 v[kPieces]; bool on[kPieces];
#pragma unroll
                            for (uint32_t j = 0; j < kPieces; ++j) {
                                const uint32_t sl = lane / kPieces + kPerInstr * j;
                                const uint32_t wbj = (uint32_t)__shfl((int)wb, (int)sl);
                                on[j] = ((lm >> sl) & 1ull) != 0ull;
                                if (on[j]) v[j] = ld16(in + wbj + piece);
                            }
#pragma unroll
                            for (uint32_t j = 0; j < kPieces; ++j) {
                                const uint32_t sl = lane / kPieces + kPerInstr * j;
                                if (on[j]) *reinterpret_cast<u32x4*>(&win[sl * kWin + piece]) = v[j];
                            }
                        }
                        uint32_t pl = wl + (p - wb);
                        // a hop needs its 4-byte view inside the window
                        const uint32_t lim = live ? wl + (stop - wb < kWin - 3u ? stop - wb : kWin - 3u) : 0u;
                        const uint32_t room = live ? (fast_end - wb < kWin ? fast_end - wb : kWin) : 0u;
#pragma unroll
                        for (uint32_t h = 0; h < (kWin - 7u) / 57u; ++h) {       // three hops need at most 57 bytes
                            if (RECORD.value) whop3_record(pl, lim, n, wl + room, wl + kWin - 4u, k, kCap + lane, wb - wl - cstart, gtoks);
                            else whop3(pl, lim, n, wl + room, wl + kWin - 4u);
                        }
                        p = wb + (pl - wl);
                        stuck = live && p == p0;
                    }
                    // the general routine takes the tokens the plain view cannot (no progress in a round) and the input's last bytes
                    const bool slow = go && p < end && p < len && (stuck || p >= stop);
                    if (!__any(slow)) { if (!__any(go && p < stop)) break; continue; }
                    LZF_COUNT(dbg_slow);
                    if (slow) {
                        uint32_t nx;
                        if (!token_next(p, nx)) { err = true; p = len; }
                        else {
                            if (RECORD.value) { gtoks[k] = (uint16_t)(p - cstart); ++k; }
                            ++n; p = nx;
                        }
                    }
                }
                return p;
            };
            // =====================================================================
            // A. speculative lane-parallel parse of one chunk: regions [cstart + i*R, +R)
            // =====================================================================
            const uint32_t rbeg = cstart + lane * (uint32_t)R;
            const uint32_t rend = rbeg + (uint32_t)R;
            uint32_t x = 0, n = 0, kdummy = 0;
            bool lerr = false;
            // First guess: walk in from kWarm bytes before the region, so that the chain has usually
            // re-synchronised by the time it enters the region (lane 0 starts at a true token).
            uint32_t start = lane == 0 ? cstart : rbeg - (kWarm < (uint32_t)R ? kWarm : (uint32_t)R);
            {
                uint32_t nw = 0; bool ew = false;
                start = walk(start, rbeg, nw, kdummy, ew, lane != 0, No{});      // warm-up: these tokens do not count
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
            // token ranks in stream order, then the record pass
            const uint32_t incl_n = wave_scan_add(n);
            const uint32_t rank0 = incl_n - n;
            const uint32_t Tc = __builtin_amdgcn_readlane(incl_n, 63);
            {
                uint32_t k = rank0, n2 = 0; bool e2 = false;
                (void)walk(start, rend, n2, k, e2, true, Yes{});
            }
            const uint32_t cend = __builtin_amdgcn_readlane(x, 63);      // where the next chunk starts
            int cerr = LZF_OK;                                           // UnexpectedEnd right after the listed tokens
            if (__ballot(lerr)) cerr = LZF_UNEXPECTED_END;
            wave_store_fence();                                          // the token list is read back below

#define LZF_TOKEN_AT(i) ((uint32_t)gtoks[(i)])
#include "../lz4_decompress_batch_phase.inc"
#undef LZF_TOKEN_AT
            if (status == LZF_OK && cerr != LZF_OK) status = cerr;
            cstart = cend;
        }
#undef RIDX
#undef PHASE
    }
    if (lane == 0) {
        results[jid].out_len = o;
        results[jid].status = status;
#ifdef LZF_DBG_COUNT
        results[jid].reserved = ((dbg_slow >> 4) > 65535u ? 65535u : (dbg_slow >> 4)) << 16 | ((dbg_rounds >> 6) > 65535u ? 65535u : (dbg_rounds >> 6));
        results[jid].out_len = (unsigned long long)o | ((unsigned long long)dbg_walks << 32);
#else
        results[jid].reserved = (uint32_t)((clock64() - t_start) >> 10);   // diagnostic: shader kilo-cycles spent on this job
#endif
    }
}

#define LZF_INSTW(NAME, RG, R_, W_) template __global__ void lzf_decompress_windowed_kernel<RG, R_, W_>(const lzf_decompress_job*, lzf_job_result*, uint32_t, uint16_t*, uint32_t);
LZF_WINDOWED_VARIANTS(LZF_INSTW)
#undef LZF_INSTW

}  // namespace lzf
