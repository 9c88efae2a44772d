// lz4_compress_compact.hip — raw::compress2 with U32Table semantics for gfx950, compact position table.
//
// Same algorithm, same speculative batches and the same output as lz4_compress.hip (see there for the
// description and the reference lines); the difference is the table.  The compress kernel is latency-
// bound (two dependent HBM/MALL round trips per sequence) and its throughput is proportional to the
// wavefronts a CU holds, which the 16 KiB of LDS a 4096 x u32 table needs caps at 10.  Here a slot keeps
// only the low 16 bits of the position; one more bit per slot (par[]) holds the parity of the position's
// 64 KiB epoch.  A candidate is acceptable only within 65535 bytes (mod.rs:201), i.e. in the current or the
// previous epoch, so those 17 bits identify it exactly — provided no entry older than that survives:
// whenever the cursor enters a new epoch E every slot whose parity equals E's (all of them from epoch
// E-2 or older) is cleared to "position (E-1) << 16", which is out of reach for the whole of epoch E
// (sweep_to; ~250 instructions per 64 KiB of input).  A batch never straddles an epoch boundary.
// The min-lane tag that finds same-slot collisions inside a batch goes through the 32-bit LDS word that
// holds two slots; the batch is cut at the first lane whose word is shared, which is either the collision
// the sequential algorithm would see (same slot) or a neighbour (then nothing happens: the cut lane is
// evaluated normally and the next batch starts behind it).  8.5 KiB of LDS per wave: 18 waves per CU.
// Jobs with a caller-owned writable table, a table offset or U16Table semantics stay on the general kernel
// (compress_job_is_compact, lzf_compress_common.h).
#include "lzf_device.h"
#include "lzf_compress_common.h"
#include "kernels.h"

namespace lzf {

namespace {
constexpr uint32_t kSlots = 4096;
constexpr uint32_t kProbeLanes = 32;     // lanes that fetch 16 input bytes for a run's first batch: its 16 probes and 16 positions behind them
// mod.rs:41-51: v = 8 bytes LE (0 if fewer than 8 remain), ((v << 24) * 889523592379) >> 52
__device__ __forceinline__ uint32_t hash5(uint64_t v8) { return (uint32_t)(((v8 << 24) * 889523592379ull) >> 52); }
// a < b as 0/1 for values below 2^31 — in vector registers on purpose: the compress kernels are bound by the CU's one scalar
// unit (hipcc keeps lane predicates as SGPR masks), so the per-sequence path does its predicate arithmetic on integers
__device__ __forceinline__ uint32_t lt01(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_sub_u32 %0, %1, %2\n\tv_lshrrev_b32 %0, 31, %0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
}  // namespace

// DRY = cost probe (capi.hip, job ordering): the same parse with every store to the output dropped; only the
// cycle count in results[].reserved is of interest.  perm (optional) maps the launch index to the job index.
template <bool DRY>
__global__ __launch_bounds__(64) void lzf_compress_compact_kernel(
    const lzf_compress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs,
    const uint32_t* __restrict__ perm, uint32_t alone) {
    // slot h = 16-bit half (h & 1) of word h >> 1 | epoch parity of slot h = bit (h & 31) of word kParBase + (h >> 5) | one
    // scratch word per lane: lanes with nothing to do aim their LDS accesses there instead of leaving the instruction (exec-mask
    // bookkeeping is scalar work, and the scalar unit is the bottleneck of this kernel)
    constexpr uint32_t kParBase = kSlots / 2, kScratch = kSlots / 2 + kSlots / 32;
#ifdef LZF_DBG_LDS_PAD
    __shared__ __attribute__((aligned(16))) uint32_t tab32[kSlots / 2 + kSlots / 32 + kWave + LZF_DBG_LDS_PAD / 4];      // occupancy experiment
#else
    __shared__ __attribute__((aligned(16))) uint32_t tab32[kSlots / 2 + kSlots / 32 + kWave];
    static_assert(sizeof(tab32) == kCompactLdsBytes, "capi.hip derives the kernel's residency from kCompactLdsBytes");
#endif
    uint32_t* const par = tab32 + kParBase;
    uint16_t* const tab16 = reinterpret_cast<uint16_t*>(tab32);
    const uint32_t tab_a = lds_addr(tab32);

    if (blockIdx.x >= n_jobs) return;
    const uint32_t jid = perm ? perm[blockIdx.x] : blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const lzf_compress_job job = jobs[jid];
    const long long t_start = clock64();
    if (!compress_job_is_compact(job)) {            // handled by lzf_compress_wave_kernel ...
        // ... unless the caller promised there are no such jobs (LZF_KINDS_U32_FRESH_ONLY) and that kernel is not launched
        if (alone && lane == 0) { results[jid].out_len = 0; results[jid].status = LZF_CONTRACT; results[jid].reserved = 0; }
        return;
    }

    cgu8* __restrict__ in = as_global(job.input);
    int status = LZF_OK;
    uint32_t work = 0;      // DRY: probe batches + sequences, the two things a block's time is made of
#ifdef LZF_DBG_PATHS        // analysis: which path each sequence / batch took; the five counts replace the first 20 output bytes
    uint32_t pc_straight = 0, pc_ext = 0, pc_tail_short = 0, pc_tail_long = 0, pc_bfast = 0, pc_bgen = 0;
#define PCOUNT(x) (++(x))
#else
#define PCOUNT(x) do { } while (0)
#endif
#ifdef LZF_PHASE_TIMING
    long long g_tph[6] = {0, 0, 0, 0, 0, 0};
#endif
    Sink s{as_global(job.out), 0u, job.out_cap > kMaxLen ? kMaxLen : (uint32_t)job.out_cap};
    // The common sequence's one store is DEFERRED to the next batch, behind the issue of its gather: a wave's loads and stores are
    // acknowledged in order, so a store issued at the end of a sequence is still on its way when the next probe bytes are waited for;
    // issued behind the gather it is the newest access at the gather's wait (vmcnt(1)) and long gone at the next one.
    uint32_t ps_n = 0, ps_pos = 0, ps_byte = 0;          // bytes pending (0: none), their place in the output, lane l's byte
    auto flush_ps = [&]() {
        if (ps_n) { if (!DRY && lane < ps_n) s.out[ps_pos + lane] = (uint8_t)ps_byte; ps_n = 0; }
    };
    {
        const uint32_t len = (uint32_t)job.input_len;
        const uint32_t init = (uint32_t)job.cursor;                      // :169
        uint32_t cursor = init;
        const uint32_t fast_lo = init + 1u > 8u ? init + 1u : 8u;       // the fast search wants c > init (:200) and 8 bytes before c
        // ... and a batch of bw probes at c inside the swept epoch with 40 readable bytes behind it: f_lo <= c && c + bw <= f_hi
        // (two scalar compares per batch; recomputed when the epoch changes)
        uint32_t f_lo = 0xFFFFFFFFu, f_hi = 0u;
        uint32_t swept = init >> 16;                                     // epoch the table is consistent with
        // ---- table in: Default::default() (:32-36) or the caller's read-only template, converted
        {
            const uint32_t e0 = swept;
            const uint32_t gone_par = e0 >= 1u ? (e0 - 1u) & 1u : 0u;    // parity that marks "out of reach" in epoch e0
            if (job.table) {
                const LZF_GLOBAL lzf_u32_table* t = (const LZF_GLOBAL lzf_u32_table*)job.table;
                for (uint32_t i = lane; i < kSlots; i += kWave) {
                    const uint32_t v = t->dict[i];
                    const uint32_t e = v >> 16;
                    const bool keep = e == e0 || e + 1u == e0;             // within reach of the first probes
                    tab16[i] = keep ? (uint16_t)v : (uint16_t)0;
                    const unsigned long long bm = __builtin_amdgcn_ballot_w64(keep ? (e & 1u) != 0u : gone_par != 0u);
                    if (lane == 0) { par[i >> 5] = (uint32_t)bm; par[(i >> 5) + 1u] = (uint32_t)(bm >> 32); }
                }
            } else {
                for (uint32_t i = lane; i < kSlots / 2; i += kWave) tab32[i] = 0u;
                for (uint32_t i = lane; i < kSlots / 32; i += kWave) par[i] = gone_par ? 0xFFFFFFFFu : 0u;
            }
        }
        auto set_fast_range = [&]() {
            const uint32_t eb = swept << 16, ee = eb + 0x10000u;            // (len < 2^31: swept < 2^15)
            f_lo = len >= 56u ? (fast_lo > eb ? fast_lo : eb) : 0xFFFFFFFFu;
            f_hi = len >= 56u ? (len - 40u < ee ? len - 40u : ee) : 0u;
        };
        set_fast_range();
        // The cursor enters epoch E (> swept): entries of epoch E-2 and older go out of reach.
        auto sweep_to = [&](uint32_t E) {
            const uint32_t keep_par = (E - 1u) & 1u;
            if (E == swept + 1u) {
                for (uint32_t w = lane; w < kSlots / 32; w += kWave) {
                    const uint32_t pw = par[w];
                    const uint32_t del = keep_par ? ~pw : pw;            // slots whose parity is E's
                    for (uint32_t t = 0; t < 16u; ++t) {
                        const uint32_t two = (del >> (2u * t)) & 3u;
                        if (two) tab32[w * 16u + t] &= ((two & 1u) ? 0u : 0xFFFFu) | ((two & 2u) ? 0u : 0xFFFF0000u);
                    }
                    par[w] = keep_par ? 0xFFFFFFFFu : 0u;
                }
            } else {                                                     // a long match skipped an epoch: nothing is in reach
                for (uint32_t i = lane; i < kSlots / 2; i += kWave) tab32[i] = 0u;
                for (uint32_t i = lane; i < kSlots / 32; i += kWave) par[i] = keep_par ? 0xFFFFFFFFu : 0u;
            }
            swept = E;
            set_fast_range();
        };

        // 8 input bytes at pos; bytes at or beyond len read as 0
        auto ld8_part = [&](uint32_t pos) -> uint64_t {
            if (pos + 8u <= len) return ld8(in + pos);
            uint64_t v = 0;
            for (uint32_t i = 0; pos + i < len; ++i) v |= (uint64_t)in[pos + i] << (8u * i);
            return v;
        };
        // The first batch of a literal run probes cursor + lane; those 16 bytes per lane are requested
        // as soon as the cursor is known (before the previous sequence is emitted) and consumed here.
        uint32_t pf_c = 0xFFFFFFFFu;
        uint64_t pfA0 = 0;        // (8 bytes per lane; lanes that were not loaded keep stale bytes: every use of a lane's hash is guarded by its being in the batch)
        uint32_t pend_q = 0xFFFFFFFFu;      // position of a `cursor - 2` insert whose bytes (pfQ, lane 16) are in flight
        uint64_t pfQ = 0;
        auto insert_hash = [&](uint32_t q, uint32_t h) {       // lane 0 writes the slot, the others their scratch word
            const uint32_t z = lane ? 1u : 0u;
            tab16[z ? 2u * (kScratch + lane) : h] = (uint16_t)q;
            const uint32_t pa = z ? kScratch + lane : kParBase + (h >> 5);
            const uint32_t bit = z ? 0u : 1u << (h & 31u);
            lds_mskor32(tab_a + 4u * pa, bit, bit & (0u - ((q >> 16) & 1u)));      // the slot's parity bit := parity of q's epoch
        };
        auto insert_at = [&](uint32_t q, uint64_t v8) { insert_hash(q, hash5(v8)); };
        // The same when lane `qi` of the wave probed position q itself, i.e. holds its hash in h: that lane alone makes the two accesses
        // (EXEC narrowed to it around them; every lane is active here, the kernel's control flow is wave-uniform) with addresses it
        // has mostly worked out for the commit already — 8 instructions instead of the 18 of a v_readlane'd hash spread back to lane 0.
        auto insert_from_lane = [&](uint32_t qi, uint32_t q, uint32_t h) {
            const uint32_t a16 = tab_a + 2u * h, pa = tab_a + 4u * (kParBase + (h >> 5)), bit = 1u << (h & 31u);
            const uint32_t val = bit & (0u - ((q >> 16) & 1u));
            unsigned long long sv;      // (the incoming EXEC saved and restored, not assumed to be all lanes: ADVICE r5)
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b16 %2, %3\n\tds_mskor_b32 %4, %5, %6\n\ts_mov_b64 exec, %0"
                         : "=&s"(sv) : "s"(1ull << qi), "v"(a16), "v"(q), "v"(pa), "v"(bit), "v"(val) : "memory");
        };
#ifdef LZF_PHASE_TIMING
        long long tq = clock64();
#define CPHASE(i) do { const long long tn = clock64(); g_tph[i] += tn - tq; tq = tn; } while (0)
#else
#define CPHASE(i) do { } while (0)
#endif

        while (cursor < len && status == LZF_OK) {                        // :171
            if (pend_q != 0xFFFFFFFFu) {
                const uint64_t v8 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(pfQ >> 32), 16) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)pfQ, 16);
                insert_at(pend_q, v8);
                pend_q = 0xFFFFFFFFu;
            }
            uint32_t ls = cursor;                                         // :172 literal_start
            uint32_t n = 0;          // probe index inside this literal run
            uint32_t c = cursor;     // position of probe n
            bool finished = false;   // last-literals path taken
            uint32_t m_pos = 0, m_cand = 0, m = 4u, bt = 0;
            bool more_m = false, more_bt = false;
            uint64_t wA0 = 0, wA1 = 0;        // the winner's 16 input bytes (general batch only: wA_valid)
            bool wA_valid = false;
            bool straight = false;            // the whole sequence was handled inside the fast search
            uint32_t ins_h = 0xFFFFFFFFu;     // fast batch: slot of the `cursor - 2` insert when its 8 bytes lie inside the winner's 16

            // ================= search, fast form for the stride-1 part of a run's schedule (its first 66 probes, mod.rs:225-231),
            // away from the block's edges and from epoch boundaries: 16 probes at cursor + lane first — most runs end there —
            // then up to 48 at a time.  Written out straight; the general batch below does the same with schedule arithmetic,
            // end-of-input lanes and epoch cuts, and takes over wherever this loop stops.
            bool found = false;
            // (a lambda so that the first batch — 16 probes, bw a compile-time constant, n == 0 — is compiled on its own)
            auto fast_batch = [&](const uint32_t bw) __attribute__((always_inline)) -> bool {    // true: the fast search is over
                if (c < f_lo) return true;
                if (c + bw > f_hi) return true;                                   // (len < 2^31: no wrap)
                if (DRY) ++work;
                PCOUNT(pc_bfast);
                const bool inb = lane < bw;
                const uint32_t ck = c + lane;
                const uint32_t have = bw > kProbeLanes ? bw : kProbeLanes;   // lanes holding 16 input bytes (c + 40 + bw <= len: readable)
                // (a run's first batch finds its probe bytes requested by the previous sequence; a scalar branch, not a per-lane select)
                if (pf_c != c) { if (lane < have) pfA0 = ld8(in + ck); }
                const uint64_t A0 = pfA0;
                pf_c = 0xFFFFFFFFu;
                const uint32_t h = hash5(A0);
                const uint32_t wi = inb ? h >> 1 : kScratch + lane;          // (lanes outside the batch: their scratch word)
                const uint32_t oldpair = tab32[wi], pw = par[h >> 5];
                tab32[wi] = kMark; atomicMin(&tab32[wi], lane);
                const uint32_t first = tab32[wi];
                // Lanes below D (the first lane whose table word is also touched by an earlier lane) are alone in their words:
                // their candidates are what the sequential code would see, and their table writes are the sequential ones.  So
                // the batch is the lanes below D: a winner among them ends the run, otherwise they commit and the next batch
                // starts at lane D's position (D >= 1).
                const uint32_t D = first_lane(__builtin_amdgcn_ballot_w64(first != lane));
                uint32_t W = 64u, cand = 0;
                {
                    const uint32_t s16 = (h & 1u) ? oldpair >> 16 : oldpair & 0xFFFFu;
                    const uint32_t ec = c >> 16, xk = ck & 0xFFFFu;
                    // (integer 0/1 arithmetic on purpose: as bools hipcc turns this into nested exec-mask control flow, ~16 scalar
                    //  instructions, and the scalar unit is what the compress kernels run out of)
                    const uint32_t diff = ((pw >> (h & 31u)) ^ ec) & 1u;              // 1: the slot's epoch is the previous one
                    cand = ((ec - diff) << 16) | s16;
                    const uint32_t dcut = D < bw ? D : bw;
                    // in the batch, cand <= ck && ck - cand <= 0xFFFF (:200-201) as ONE unsigned compare: a candidate behind ck wraps to a
                    // huge distance.  (ec == 0 with diff == 1 — "the epoch before the first" — cannot occur: in epoch 0 every parity bit is 0,
                    // table-in and sweep_to above.)
                    const bool reach = lane < dcut && ck - cand <= 0xFFFFu;
                    // candidate side: the 4 bytes of the accept test (:204-206) and nothing else — the match is measured by the whole
                    // wave once the winner is known (below), not worked out by every lane for its own candidate
                    uint32_t B4 = ~(uint32_t)A0;      // (a lane without a candidate in reach: bytes that cannot match)
                    if (reach) B4 = ld4(in + cand);
                    flush_ps();                       // (the previous sequence's store: behind this batch's gather)
                    W = first_lane(__builtin_amdgcn_uicmp(B4, (uint32_t)A0, 32 /* eq */));      // :204-206 (m >= 4); the compare's mask itself
                    CPHASE(0);
                    {
                        // commit: the lanes up to the winner, or all below the cut, write their position; the others restore their
                        // word; a shared word is written by its first lane only (the later ones lie behind the cut)
                        const uint32_t cut = W < 64u ? W + 1u : dcut;
                        const uint32_t newpair = (h & 1u) ? (oldpair & 0xFFFFu) | (xk << 16) : (oldpair & 0xFFFF0000u) | xk;
                        const uint32_t keep = lt01(lane, cut);
                        tab32[first == lane ? wi : kScratch + lane] = keep ? newpair : oldpair;
                        const uint32_t pa = keep ? kParBase + (h >> 5) : kScratch + lane;
                        const uint32_t bit = (1u << (h & 31u)) & (0u - keep);
                        lds_mskor32(tab_a + 4u * pa, bit, bit & (0u - (ec & 1u)));
                        if (W >= 64u) { n += cut; c += cut; }                      // these probes advance by 1
                    }
                }
                if (W < 64u) {
                    // The match, measured by the wave (round 5, from the team kernel): lanes 0..31 compare bytes 4..35 forwards (:203-204;
                    // here len - ck >= 56: the bound :195 cannot cut them), lanes 32..63 bytes 1..32 backwards, bounded by the run and the
                    // input's start (:211-212) — two byte loads per lane from lines the gather and the probes just touched, one ballot,
                    // two s_ff1, instead of 24 gathered bytes per lane and ~60 instructions of 64-bit arithmetic, packing and unpacking.
                    m_pos = c + W;
                    m_cand = __builtin_amdgcn_readlane(cand, W);
                    const uint32_t runl = m_pos - ls, maxbt = runl < m_cand ? runl : m_cand;
                    const uint32_t li = lane & 31u;
                    const bool bwd = lane >= 32u, okb = li < maxbt;
                    const uint32_t oa = bwd ? (okb ? 0u - 1u - li : 0u) : 4u + li;          // (a backward lane beyond the bound reads a harmless byte)
                    const uint32_t xa = in[m_pos + oa], xb = in[m_cand + oa];
                    const unsigned long long mk = __builtin_amdgcn_uicmp(xa, xb, 33 /* ne */) | (0xFFFFFFFF00000000ull & ~__builtin_amdgcn_uicmp(li, maxbt, 36 /* ult */));
                    const uint32_t mlo = (uint32_t)mk, mhi = (uint32_t)(mk >> 32);
                    const uint32_t wm = mlo ? 4u + (uint32_t)__builtin_ctz(mlo) : 36u, wbt = mhi ? (uint32_t)__builtin_ctz(mhi) : 32u;
                    // table.replace(input, cursor - 2) (:218): the 8 bytes at cursor - 2 are the probe bytes of lane W + m - 2, whose
                    // hash is already there (lanes up to `have` hold probe bytes; an extended match is handled at the insert)
                    const uint32_t qi = W + wm - 2u;
                    found = true;
                    // The common sequence in one straight line (everything the general tail below decides with a branch each is
                    // known here): no extension (m < 16, no more backtrack), the insert's hash at hand, no epoch change and 56
                    // readable bytes at the new cursor, <= 57 literals (one LSIC byte at most), match length in the token, room in the sink —
                    // the last six as one sign test (every term is negative when its condition fails; all quantities are below 2^31).
                    const uint32_t cur2 = m_pos + wm, ex2 = wm - 4u + wbt, L2 = (m_pos - wbt) - ls;
                    const uint32_t nl2 = L2 >= 15u ? 1u : 0u;                          // literal length beyond the token: one LSIC byte up to 57
                    const int32_t inrange = (int32_t)(have - 1u - qi) | (int32_t)(f_hi - kFirstBatch - cur2) | (int32_t)(14u - ex2) |
                                            (int32_t)(57u - L2) | (int32_t)(s.cap - s.pos - (L2 + 3u + nl2));
                    if (mlo != 0u && mhi != 0u && inrange >= 0) {
                        if (DRY) ++work;
                        cursor = cur2;                                                 // :215
                        // lane j: literal j - 1 - nl2.  A run that ends in its first batch has them in registers — lane k probed
                        // position ls + k, the low byte of its A0 is literal k — one or two lanes to the left (round 4: the byte load this
                        // replaces was one more access for the next probe bytes' wait to include; vmcnt counts in order)
                        uint32_t byte;
                        if (n == 0u) {
                            const uint32_t b1 = wave_prev((uint32_t)A0 & 0xFFu, 0u), b2 = wave_prev(b1, 0u);
                            byte = nl2 ? b2 : b1;
                        } else {
                            const uint32_t lt = lane > nl2 ? lane - nl2 : 0u, lj = lt < L2 ? lt : L2;
                            byte = in[ls + (lj ? lj - 1u : 0u)];
                        }
                        if (lane < kProbeLanes) pfA0 = ld8(in + cur2 + lane);          // the next run's probe bytes, right behind
                        pf_c = cur2;
                        insert_from_lane(qi, cur2 - 2u, h);                            // :218 (lane qi probed cursor - 2: have - 1 - qi >= 0 above)
                        const uint32_t off2 = m_pos - m_cand;                          // :208
                        if (lane == 1u && nl2) byte = L2 - 15u;                        // write_group, :150-163
                        if (lane == 0u) byte = ((nl2 ? 15u : L2) << 4) | ex2;
                        if (lane == L2 + nl2 + 1u) byte = off2;
                        if (lane == L2 + nl2 + 2u) byte = off2 >> 8;
                        ps_byte = byte; ps_pos = s.pos; ps_n = L2 + nl2 + 3u;           // (stored behind the next batch's gather)
                        s.pos += L2 + nl2 + 3u;
                        PCOUNT(pc_straight);
                        straight = true;
                    } else {
                        m = wm;
                        more_m = mlo == 0u;
                        bt = wbt;
                        more_bt = mhi == 0u;
                        if (!more_m && qi < have) ins_h = __builtin_amdgcn_readlane(h, qi);
                    }
                    return true;
                }
                return false;
            };
            // The straight sequences of consecutive runs as a loop of their own (round 5): its only loop-carried state is the cursor, the
            // sink, the deferred store and the prefetched probe bytes — as part of the big loop every iteration paid a dozen scalar moves
            // that merge its control flow with the rare paths'.  (A straight sequence leaves cursor + 16 <= f_hi, no pending insert.)
            bool fast_over;
            for (;;) {
                fast_over = fast_batch(kFirstBatch);
                if (!straight) break;
                straight = false; found = false;
                ls = cursor; c = cursor; n = 0;
            }
            if (!fast_over)
                while (n < 58u) { if (fast_batch(66u - n < 48u ? 66u - n : 48u)) break; }
            if (straight) continue;
            // A match of the fast search that the straight line above did not finish — longer than the token holds (a length tail,
            // mod.rs:243-260), or all 32 measured bytes equal (m == 36 so far) and ending within the next 512 bytes — with a short
            // literal run, also goes in a straight line: if needed one compare round of 8 bytes per lane (:203-204), the length's LSIC
            // tail in the same store, the `cursor - 2` insert deferred to the top of the next iteration like in the general tail.
            // (Kept out of the search loop: inside it, it costs the common path scalar moves.)
            if (found && !more_bt && (!more_m || (m == 36u && m_pos + 36u + 512u + 5u <= len))) {
                const uint32_t L3 = (m_pos - bt) - ls;
                if (L3 < 15u) {
                    uint32_t me = m;
                    bool have_me = !more_m;
                    if (more_m) {
                        const uint64_t x = ld8(in + m_pos + 36u + lane * 8u) ^ ld8(in + m_cand + 36u + lane * 8u);
                        const unsigned long long neq = __builtin_amdgcn_ballot_w64(x != 0ull);
                        if (neq) {
                            const uint32_t fl = (uint32_t)__builtin_ctzll(neq);
                            const uint32_t xlo = __builtin_amdgcn_readlane((uint32_t)x, fl), xhi = __builtin_amdgcn_readlane((uint32_t)(x >> 32), fl);
                            me = 36u + fl * 8u + (uint32_t)(__builtin_ctzll(((uint64_t)xhi << 32) | xlo) >> 3);
                            have_me = true;
                        }
                    }
                    if (have_me) {
                        const uint32_t cur3 = m_pos + me, ex3 = me - 4u + bt;                          // ex3 in 0 .. 575
                        const uint32_t nt = ex3 < 15u ? 0u : 1u + (ex3 >= 270u ? 1u : 0u) + (ex3 >= 525u ? 1u : 0u);   // lsic_len
                        const uint32_t tot3 = L3 + 3u + nt;
                        if (cur3 + kFirstBatch <= f_hi && s.cap - s.pos >= tot3) {
                            if (DRY) ++work;
                            cursor = cur3;                                                 // :215
                            const uint32_t lj = lane < L3 ? lane : L3;
                            uint32_t byte = in[ls + (lj ? lj - 1u : 0u)];                // lane j: literal j-1
                            if (lane < kProbeLanes) pfA0 = ld8(in + cur3 + lane);
                            pf_c = cur3;
                            if (lane == 16u) pfQ = ld8(in + cur3 - 2u);                   // :218, inserted before the next probes
                            pend_q = cur3 - 2u;
                            const uint32_t off3 = m_pos - m_cand;                          // :208
                            if (lane > L3 + 2u) byte = 0xFFu;                              // LSIC tail: 0xFF ..., then the rest
                            if (lane + 1u == tot3 && nt) byte = ex3 - 15u - 255u * (nt - 1u);
                            if (lane == 0u) byte = (L3 << 4) | (ex3 < 15u ? ex3 : 15u);
                            if (lane == L3 + 1u) byte = off3;
                            if (lane == L3 + 2u) byte = off3 >> 8;
                            flush_ps();
                            ps_byte = byte; ps_pos = s.pos; ps_n = tot3;                   // (deferred like the common sequence's)
                            s.pos += tot3;
                            PCOUNT(pc_ext);
                            continue;
                        }
                    }
                }
            }
            // ================= search: speculative batches of the :177-232 loop
            if (!found) for (;;) {
                flush_ps();
                if (DRY) ++work;
                PCOUNT(pc_bgen);
                { const uint32_t eb = c >> 16; if (eb != swept) sweep_to(eb); }        // the batch base enters a new 64 KiB epoch
                // Common case, decided once per batch with scalar compares: first batch of a run, not at
                // the block's edges.  Then every lane is a plain probe (no schedule arithmetic, no end-of-
                // input lanes, full 16-byte loads in range, positions fit the slot type).
                const bool easy = n == 0u && c > init && c >= 8u && c + kFirstBatch + 40u <= len &&
                                  ((c + kFirstBatch - 1u) >> 16) == (c >> 16);          // and the batch stays inside one epoch
                // probe positions: the first 66 probes of a run advance by 1 (mod.rs:225-231)
                uint32_t ck, sn = 0;
                if (n + 64u <= 66u) ck = c + lane;
                else { sn = sched_prefix(n); ck = c + (sched_prefix(n + lane) - sn); }
                // A run's first batch is narrow: on compressible data the match is almost always among the
                // first probes, and 16 scattered table/candidate accesses cost far less than 64 (LDS bank
                // conflicts, one cache line per lane); the batch widens once the run has missed 16 times.
                const uint32_t bw = n == 0u ? kFirstBatch : kWave;
                const bool inb = lane < bw;
                const bool endk = easy ? false : (inb && ((ck > len) || (len - ck < 12u)));   // :178
                const bool epk = easy ? false : (inb && (ck >> 16) != (c >> 16));     // beyond the epoch of the batch base: next batch
                const bool active = inb && !endk && !epk;
                uint64_t A0 = 0, A1 = 0;                                  // input[ck .. ck+16)
                if (n == 0u && pf_c == c) { A0 = pfA0; if (easy) { if (inb) A1 = ld8(in + ck + 8u); } else if (active) A1 = ld8_part(ck + 8u); }
                else if (easy) { if (inb) { A0 = ld8(in + ck); A1 = ld8(in + ck + 8u); } }
                else if (active) { A0 = ld8(in + ck); A1 = ld8_part(ck + 8u); }   // >= 12 bytes remain
                pf_c = 0xFFFFFFFFu;
                const uint32_t h = hash5(A0);
                const uint32_t wi = h >> 1;                               // two 16-bit slots per LDS word
                uint32_t oldpair = 0, pw = 0, first = lane;
                if (active) oldpair = tab32[wi];
                if (active) pw = par[h >> 5];
                if (active) tab32[wi] = kMark;                            // min-lane tag through the word the slot lives in
                if (active) atomicMin(&tab32[wi], lane);
                if (active) first = tab32[wi];
                const bool conf = active && first != lane;                // an earlier lane touches the same word
                const uint32_t D = first_lane(__builtin_amdgcn_ballot_w64(conf));            // 64 = none; the batch is cut after lane D
                const uint32_t e_end = easy ? 64u : first_lane(__builtin_amdgcn_ballot_w64(endk));
                const uint32_t x_end = easy ? 64u : first_lane(__builtin_amdgcn_ballot_w64(epk));
                // lanes below D are alone in their word, so lane D's word holds exactly one earlier lane, fD: either the same
                // slot (the collision the sequential algorithm would see) or the neighbouring slot (nothing to see)
                uint32_t fD = 64u; bool true_dup = false;
                if (D < 64u) {
                    fD = __builtin_amdgcn_readlane(first, D) & 63u;
                    true_dup = __builtin_amdgcn_readlane(h, D) == __builtin_amdgcn_readlane(h, fD);
                }
                // candidate the sequential algorithm would see at lane k (k <= D): the slot holds the low 16 bits of a
                // position and the parity of its epoch; epochs older than the previous one were swept (sweep_to)
                uint32_t cand; bool inwin;
                {
                    const uint32_t s16 = (h & 1u) ? oldpair >> 16 : oldpair & 0xFFFFu;
                    const uint32_t pb = (pw >> (h & 31u)) & 1u;
                    const uint32_t ec = ck >> 16, xk = ck & 0xFFFFu;
                    const bool same = pb == (ec & 1u);
                    cand = ((same ? ec : ec - 1u) << 16) | s16;
                    inwin = same ? s16 <= xk : (ec >= 1u && s16 > xk);   // <=> cand <= ck && ck - cand <= 0xFFFF (:200-201)
                    if (true_dup && lane == D) {
                        cand = __builtin_amdgcn_readlane(ck, fD);
                        inwin = ck - cand <= 0xFFFFu;
                    }
                }
                // candidate side, one round trip: 16 bytes at the candidate for the >= 4 test and the
                // forward extension, 8 bytes before both positions for the backtrack
                const bool reach = active && lane <= D && ck != init && inwin;
                uint64_t B0 = 0, B1 = 0, PA = 0, PB = 0;
                const bool btfast = cand >= 8u;        // (then ck >= 8 as well)
                if (reach) {
                    B0 = ld8(in + cand);
                    if (easy) B1 = ld8(in + cand + 8u); else B1 = ld8_part(cand + 8u);
                    if (btfast) { PA = ld8(in + ck - 8u); PB = ld8(in + cand - 8u); }
                }
                const bool valid = reach && (uint32_t)A0 == (uint32_t)B0;  // :204-206 (m >= 4)
                uint32_t m_loc = 0, bt_loc = 0, maxbt = 0;
                if (valid) {
                    const uint64_t x0 = A0 ^ B0, x1 = A1 ^ B1;
                    m_loc = x0 ? (uint32_t)(__builtin_ctzll(x0) >> 3) : 8u + (x1 ? (uint32_t)(__builtin_ctzll(x1) >> 3) : 8u);
                    const uint32_t runlen = ck - ls;
                    maxbt = runlen < cand ? runlen : cand;                 // :211-212 bounds
                    if (btfast) {
                        const uint64_t xp = PA ^ PB;
                        bt_loc = xp ? (uint32_t)(__builtin_clzll(xp) >> 3) : 8u;
                    }
                }
                const uint32_t W = first_lane(__builtin_amdgcn_ballot_w64(valid));
                // last lane whose `replace` really executed in sequential order (+1)
                uint32_t commit_end;   // lanes [0, commit_end) commit
                int outcome;           // 0 = continue, 1 = match at W, 2 = end of input
                if (W < e_end && W <= D) { commit_end = W + 1u; outcome = 1; }
                else if (D < 64u) { commit_end = D + 1u; outcome = 0; }
                else if (e_end < 64u && e_end <= x_end) { commit_end = e_end; outcome = 2; }
                else if (x_end < 64u) { commit_end = x_end; outcome = 0; }
                else { commit_end = bw; outcome = 0; }
                // ---- commit: restore the tagged words, then write the positions the sequential code would have written
                {
                    const bool overridden = true_dup && D < commit_end && lane == fD;          // lane D writes that slot
                    const bool commits = active && lane < commit_end && lane != D && !overridden;
                    const uint32_t xk = ck & 0xFFFFu;
                    const uint32_t newpair = (h & 1u) ? (oldpair & 0xFFFFu) | (xk << 16) : (oldpair & 0xFFFF0000u) | xk;
                    if (active && !commits && lane != D) tab32[wi] = oldpair;                   // (lane D's word belongs to fD)
                    if (commits) tab32[wi] = newpair;
                    const bool dwrites = active && lane == D && D < commit_end;
                    if (dwrites) tab16[h] = (uint16_t)xk;
                    if (commits || dwrites) {
                        const uint32_t bit = 1u << (h & 31u);
                        lds_mskor32(tab_a + 4u * (kParBase + (h >> 5)), bit, bit & (0u - ((c >> 16) & 1u)));
                    }
                }
                if (status != LZF_OK) break;
                if (outcome == 1) {
                    m_pos = __builtin_amdgcn_readlane(ck, W);
                    m_cand = __builtin_amdgcn_readlane(cand, W);
                    const uint32_t alen_w = (len - 5u) - m_pos;            // :195
                    const uint32_t mw = __builtin_amdgcn_readlane(m_loc, W);
                    m = mw < alen_w ? mw : alen_w;
                    more_m = mw >= 16u && alen_w > 16u;
                    const uint32_t mbw = __builtin_amdgcn_readlane(maxbt, W);
                    const uint32_t btw = __builtin_amdgcn_readlane(bt_loc, W);
                    const bool fastw = m_cand >= 8u;
                    bt = fastw ? (btw < mbw ? btw : mbw) : 0u;
                    more_bt = fastw ? (btw >= 8u && mbw > 8u) : (mbw > 0u);
                    wA0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(A0 >> 32), W) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)A0, W);
                    wA1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(A1 >> 32), W) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)A1, W);
                    wA_valid = true;
                    break;
                }
                if (outcome == 2) { finished = true; break; }
                n += commit_end;
                if (n <= 66u) c += commit_end;
                else c += sched_prefix(n) - (sn ? sn : sched_prefix(n - commit_end));
            }
            if (status != LZF_OK) break;
            CPHASE(1);

            if (finished) {
                flush_ps();
                // ---- last literals, mod.rs:178-190
                const uint32_t L = len - ls;
                const uint32_t nl = lsic_len(L);
                const uint32_t total = 1u + nl + L;
                if (s.cap - s.pos < total) {
                    // which individual write fails does not matter: the sink content is dropped
                    status = LZF_OUTPUT_FULL;
                    break;
                }
                if (!DRY) {
                    gu8* d = s.out + s.pos;
                    if (lane == 0) d[0] = (uint8_t)((L < 15u ? L : 15u) << 4);
                    if (nl) lsic_store(d + 1, L, nl, lane);
                    wave_copy(d + 1u + nl, in + ls, L, lane);
                }
                s.pos += total;
                cursor = len;
                break;
            }

            // ================= match found at m_pos against m_cand (distance checked)
            // forward extension beyond the 16 bytes compared in registers (:195,:203-204)
            const uint32_t alen = (len - 5u) - m_pos;
            if (more_m) {
                cgu8* a = in + m_pos;
                cgu8* b = in + m_cand;
                bool done = false;
                while (!done && alen - m >= 512u) {              // 8 bytes per lane
                    const uint64_t x = ld8(a + m + lane * 8u) ^ ld8(b + m + lane * 8u);
                    const unsigned long long neq = __builtin_amdgcn_ballot_w64(x != 0ull);
                    if (neq) {
                        const uint32_t fl = (uint32_t)__builtin_ctzll(neq);
                        const uint32_t xlo = __builtin_amdgcn_readlane((uint32_t)x, fl), xhi = __builtin_amdgcn_readlane((uint32_t)(x >> 32), fl);
                        const uint64_t xf = ((uint64_t)xhi << 32) | xlo;
                        m += fl * 8u + (uint32_t)(__builtin_ctzll(xf) >> 3);
                        done = true;
                    } else {
                        m += 512u;
                    }
                }
                while (!done) {                                   // 1 byte per lane
                    const uint32_t i = m + lane;
                    const bool inr = i < alen;
                    bool ne = true;
                    if (inr) ne = a[i] != b[i];
                    const unsigned long long neq = __builtin_amdgcn_ballot_w64(ne);   // out-of-range lanes stop the scan
                    if (neq) { m += (uint32_t)__builtin_ctzll(neq); done = true; }
                    else m += 64u;
                }
            }
            // backtrack beyond the 8 bytes compared in registers (:211-212)
            if (more_bt) {
                const uint32_t maxbt = (m_pos - ls) < m_cand ? (m_pos - ls) : m_cand;
                bool done = false;
                while (!done) {
                    const uint32_t i = bt + lane;
                    bool ne = true;
                    if (i < maxbt) ne = in[m_pos - 1u - i] != in[m_cand - 1u - i];
                    const unsigned long long neq = __builtin_amdgcn_ballot_w64(ne);
                    if (neq) { bt += (uint32_t)__builtin_ctzll(neq); done = true; }
                    else bt += 64u;
                }
            }
            cursor = m_pos + m;                                            // :215
            if (DRY) ++work;
            if ((cursor >> 16) != swept) sweep_to(cursor >> 16);
            CPHASE(1);
            // The literal run is loaded first and the next run's first probes are requested right behind it, so
            // both travel in one round trip (loads return in order: storing the literals then waits for the
            // literal loads only, and the probes land while this sequence is emitted).
            const uint32_t lit_len = (m_pos - bt) - ls;                    // = L below
            const bool lit_fast = lit_len <= 1024u;
            uint32_t lit_b = 0, lit_t = 0; u32x4 lit_v = {0, 0, 0, 0};
            if (lit_fast) {
                if (lit_len < kWave) {                                            // lane j holds literal j-1 (the other lanes: some literal)
                    const uint32_t lj = lane < lit_len ? lane : lit_len;
                    lit_b = in[ls + (lj ? lj - 1u : 0u)];                        // (ls < len: a readable byte even without literals)
                }
                else {
                    const uint32_t bulk = lit_len & ~15u;
                    if (lane * 16u < bulk) lit_v = ld16(in + ls + lane * 16u);
                    if (lane < lit_len - bulk) lit_t = in[ls + bulk + lane];
                }
            }
            {
                const uint32_t ckn = cursor + lane;
                if (cursor + kFirstBatch + 40u <= len) { if (lane < kProbeLanes) pfA0 = ld8(in + ckn); }   // (16 probes + the 16 positions behind them: the `cursor - 2` insert's hash)
                else if (lane < kFirstBatch && ckn <= len && len - ckn >= 12u) pfA0 = ld8(in + ckn);
                pf_c = cursor;
            }
            // table.replace(input, cursor - 2) — unconditional (:218, quirks B1/B3).  When the 8 bytes at cursor - 2 are
            // not in the winner's registers they are requested with the probes (lane 16) and the insert is made at the top
            // of the next iteration, before anything reads the table — no round trip of its own.
            {
                const uint32_t q = cursor - 2u;
                if (ins_h != 0xFFFFFFFFu) insert_hash(q, ins_h);            // worked out by the winner's lane
                else {
                    bool now = true; uint64_t v8 = 0;
                    if (len - q >= 8u) {                                       // :43: fewer than 8 bytes left -> 0
                        if (wA_valid && m - 2u + 8u <= 16u) {                              // still inside the (general batch) winner's 16 bytes
                            const uint32_t sh = (m - 2u) * 8u;
                            v8 = sh == 0u ? wA0 : sh < 64u ? ((wA0 >> sh) | (wA1 << (64u - sh))) : (wA1 >> (sh - 64u));
                        } else {
                            now = false;
                            if (lane == 16u) pfQ = ld8(in + q);
                            pend_q = q;
                        }
                    }
                    if (now) insert_at(q, v8);
                }
            }
            const uint32_t dup_offset = m_pos - m_cand;                    // :208
            const uint32_t extra = m - 4u + bt;                            // :206,:214
            CPHASE(2);
            // ================= write_group, mod.rs:150-163 (+ :235 literal slice)
            const uint32_t lit_end = cursor - extra - 4u;
            const uint32_t L = lit_end - ls;
            flush_ps();
            if (L < 15u && extra < 15u) {
                // the common sequence: token, up to 14 literals, offset — one byte per lane, one store
                const uint32_t total = L + 3u;
                if (s.cap - s.pos < total) { status = LZF_OUTPUT_FULL; break; }
                uint32_t byte = lit_b;
                if (lane == 0u) byte = (L << 4) | extra;
                if (lane == L + 1u) byte = dup_offset;
                if (lane == L + 2u) byte = dup_offset >> 8;
                ps_byte = byte; ps_pos = s.pos; ps_n = total;                              // (deferred like the straight path's; flush_ps() ran above)
                s.pos += total;
                PCOUNT(pc_tail_short);
                CPHASE(3);
                continue;
            }
            PCOUNT(pc_tail_long);
            const uint32_t nl = lsic_len(L), ne = lsic_len(extra);
            const uint32_t total = 1u + nl + L + 2u + ne;
            if (s.cap - s.pos < total) { status = LZF_OUTPUT_FULL; break; }
            if (!DRY) {
            gu8* d = s.out + s.pos;
            if (lane == 0) {
                d[0] = (uint8_t)(((L < 15u ? L : 15u) << 4) | (extra < 15u ? extra : 15u));
                d[1u + nl + L] = (uint8_t)dup_offset;
                d[2u + nl + L] = (uint8_t)(dup_offset >> 8);
            }
            if (nl) lsic_store(d + 1, L, nl, lane);
            if (lit_fast) {
                gu8* ld = d + 1u + nl;
                if (L < kWave) { if (lane >= 1u && lane <= L) ld[lane - 1u] = (uint8_t)lit_b; }
                else {
                    const uint32_t bulk = L & ~15u;
                    if (lane * 16u < bulk) st16(ld + lane * 16u, lit_v);
                    if (lane < L - bulk) ld[bulk + lane] = (uint8_t)lit_t;
                }
            } else {
                wave_copy(d + 1u + nl, in + ls, L, lane);
            }
            if (ne) lsic_store(d + 3u + nl + L, extra, ne, lane);
            }
            s.pos += total;
            CPHASE(3);
        }
    }
    flush_ps();
#ifdef LZF_DBG_PATHS
    if (lane == 0 && job.out_cap >= 24u) {
        LZF_GLOBAL uint32_t* pc = (LZF_GLOBAL uint32_t*)as_global(job.out);
        pc[0] = pc_straight; pc[1] = pc_ext; pc[2] = pc_tail_short; pc[3] = pc_tail_long; pc[4] = pc_bfast; pc[5] = pc_bgen;
    }
#endif
    if (lane == 0) {
        results[jid].out_len = s.pos;
        results[jid].status = status;
#ifdef LZF_PHASE_TIMING
        { uint32_t pk = 0; for (int i = 0; i < 4; ++i) { uint32_t u = (uint32_t)(g_tph[i] >> 23); if (u > 255u) u = 255u; pk |= u << (8 * i); } results[jid].reserved = pk; }
#else
        results[jid].reserved = DRY ? work : (uint32_t)((clock64() - t_start) >> 10);   // diagnostic: shader kilo-cycles spent on this job
#endif
    }
}

template __global__ void lzf_compress_compact_kernel<false>(const lzf_compress_job*, lzf_job_result*, uint32_t, const uint32_t*, uint32_t);
template __global__ void lzf_compress_compact_kernel<true>(const lzf_compress_job*, lzf_job_result*, uint32_t, const uint32_t*, uint32_t);

}  // namespace lzf
