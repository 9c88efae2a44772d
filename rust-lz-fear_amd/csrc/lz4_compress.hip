// lz4_compress.hip — batched raw::compress2 for gfx950 (MI355X), wave64, bit-exact.
//
// Replaces src/raw/compress/mod.rs:165-238 of lz-fear (compress2) together with its tables
// (:27-101), count_matching_bytes (:117-145) and write_group / LSIC coding (:150-163,:239-260)
// for many blocks per launch.  One wavefront compresses one block; the position table lives
// in LDS (16 KiB for U32Table, 32 KiB for the U16Table variant).
//
// The reference parse is a strictly sequential state machine (greedy first match, single-slot
// table, skip schedule); its output depends on the exact order of table updates.  The kernel
// reproduces it with *speculative batches*: the 64 lanes evaluate the next 64 probe positions
// of the skip schedule (closed form below) as if every earlier probe in the batch missed, and
// the wave then commits exactly the prefix the sequential algorithm would have executed:
//   1. lane k hashes input[c_k..c_k+8) and reads its table slot (the pre-batch value);
//   2. same-slot collisions inside the batch are found by a min-lane tag written through the
//      slot itself (ds_write MARK, ds_min lane, ds_read): lane k is "first" for its slot iff
//      the tag equals k.  The batch is cut after the first lane D that is not first — its
//      candidate is the position of the (unique) earlier lane with the same slot;
//   3. the winner W is the first lane whose candidate passes the reference's accept test
//      (mod.rs:200-206), or the first lane that hits the <12-bytes-left rule (:178);
//   4. lanes <= W write their positions (the sequential `replace` calls that really happened),
//      every other touched slot gets its pre-batch value back.
// Match extension and backtracking (:204,:211-212) are wave-parallel compares + ballot.
#include "lzf_device.h"
#include "lzf_compress_common.h"

namespace lzf {

template <int KIND> struct TableTraits;
template <> struct TableTraits<LZF_TABLE_U32> {
    static constexpr uint32_t kSlots = 4096;
    static constexpr uint64_t kLimit = 0xFFFFFFFFull;                 // mod.rs:75
    // mod.rs:41-51: v = 8 bytes LE (0 if fewer than 8 remain), ((v << 24) * 889523592379) >> 52
    static __device__ __forceinline__ uint32_t hash(uint64_t v8) {
        return (uint32_t)(((v8 << 24) * 889523592379ull) >> 52);
    }
};
template <> struct TableTraits<LZF_TABLE_U16> {
    static constexpr uint32_t kSlots = 8192;
    static constexpr uint64_t kLimit = 0xFFFFull;                     // mod.rs:100
    // mod.rs:58-61: (u32 * 2654435761) >> 19
    static __device__ __forceinline__ uint32_t hash(uint64_t v8) {
        return ((uint32_t)v8 * 2654435761u) >> 19;
    }
};

template <int KIND>
__global__ __launch_bounds__(64) void lzf_compress_wave_kernel(
    const lzf_compress_job* __restrict__ jobs, lzf_job_result* __restrict__ results, uint32_t n_jobs, uint32_t skip_compact,
    const uint32_t* __restrict__ perm) {
    using TT = TableTraits<KIND>;
    __shared__ uint32_t tab[TT::kSlots];
#ifdef LZF_DBG_LDS_PAD
    __shared__ uint32_t dbg_pad[LZF_DBG_LDS_PAD / 4];     // occupancy experiment
    if (threadIdx.x == 999) dbg_pad[0] = 1;
#endif

    if (blockIdx.x >= n_jobs) return;
    const uint32_t jid = perm ? perm[blockIdx.x] : blockIdx.x;      // launch index -> job (capi.hip: longest jobs first)
    const uint32_t lane = threadIdx.x;
    const lzf_compress_job job = jobs[jid];
    const long long t_start = clock64();
    if (job.table_kind != (uint32_t)KIND) return;   // handled by the other instantiation
    if (skip_compact == 1u && compress_job_is_compact(job)) return;   // handled by lzf_compress_compact_kernel
    if (skip_compact == 2u && compress_job_is_team(job) && results[jid].status != kTeamRetry) return;   // handled by lzf_compress_team_kernel (unless it hands the job back)

    cgu8* __restrict__ in = as_global(job.input);
    int status = LZF_OK;
#ifdef LZF_PHASE_TIMING
    long long g_tph[4] = {0, 0, 0, 0};
#endif
    Sink s{as_global(job.out), 0u, job.out_cap > kMaxLen ? kMaxLen : (uint32_t)job.out_cap};
    uint64_t base_off = 0;   // EncoderTable.offset (mod.rs:30,:81)

    // ---- table in: Default::default() (:32-36) or the caller's table
    if (job.table) {
        if (KIND == LZF_TABLE_U32) {
            const LZF_GLOBAL lzf_u32_table* t = (const LZF_GLOBAL lzf_u32_table*)job.table;
            for (uint32_t i = lane; i < TT::kSlots; i += kWave) tab[i] = t->dict[i];
            base_off = t->offset;
        } else {
            const LZF_GLOBAL lzf_u16_table* t = (const LZF_GLOBAL lzf_u16_table*)job.table;
            for (uint32_t i = lane; i < TT::kSlots; i += kWave) tab[i] = t->dict[i];
            base_off = t->offset;
        }
    } else {
        for (uint32_t i = lane; i < TT::kSlots; i += kWave) tab[i] = 0u;
    }

    if (job.input_len > TT::kLimit || job.input_len >= kMaxLen) {
        status = LZF_CONTRACT;                                           // mod.rs:167
    } else {
        const uint32_t len = (uint32_t)job.input_len;
        const uint32_t init = job.cursor > job.input_len ? len : (uint32_t)job.cursor;   // :169 (a cursor past the end: the loop at :171 never runs)
        uint32_t cursor = init;
        const uint32_t boff = (uint32_t)base_off;   // low 32 bits; overflow is checked at commit

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
        uint64_t pfA0 = 0, pfA1 = 0;
#ifdef LZF_PHASE_TIMING
        long long tq = clock64();
#define CPHASE(i) do { const long long tn = clock64(); g_tph[i] += tn - tq; tq = tn; } while (0)
#else
#define CPHASE(i) do { } while (0)
#endif

        while (cursor < len && status == LZF_OK) {                        // :171
            const uint32_t ls = cursor;                                   // :172 literal_start
            uint32_t n = 0;          // probe index inside this literal run
            uint32_t c = cursor;     // position of probe n
            bool finished = false;   // last-literals path taken
            uint32_t m_pos = 0, m_cand = 0, m = 4u, bt = 0;
            bool more_m = false, more_bt = false;
            uint64_t wA0 = 0, wA1 = 0;        // the winner's 16 input bytes

            // ================= search: speculative batches of the :177-232 loop
            for (;;) {
                // Common case, decided once per batch with scalar compares: first batch of a run, not at
                // the block's edges.  Then every lane is a plain probe (no schedule arithmetic, no end-of-
                // input lanes, full 16-byte loads in range, positions fit the slot type).
                const bool easy = n == 0u && c > init && c >= 8u && (uint64_t)c + kFirstBatch + 40u <= len &&
                                  (uint64_t)c + kFirstBatch + base_off <= TT::kLimit;
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
                const bool active = inb && !endk;
                uint64_t A0 = 0, A1 = 0;                                  // input[ck .. ck+16)
                if (n == 0u && pf_c == c) { A0 = pfA0; A1 = pfA1; }
                else if (easy) { if (inb) { A0 = ld8(in + ck); A1 = ld8(in + ck + 8u); } }
                else if (active) { A0 = ld8(in + ck); A1 = ld8_part(ck + 8u); }   // >= 12 bytes remain
                pf_c = 0xFFFFFFFFu;
                const uint32_t h = TT::hash(A0);
                uint32_t old = 0, first = lane;
                if (active) old = tab[h];
                if (active) tab[h] = kMark;
                if (active) atomicMin(&tab[h], lane);
                if (active) first = tab[h];
                const bool dup = active && first != lane;
                const uint32_t D = first_lane(__ballot(dup));             // 64 = no collision
                const uint32_t e_end = easy ? 64u : first_lane(__ballot(endk));
                // candidate the sequential algorithm would see at lane k (k <= D)
                uint32_t cand;
                {
                    const uint64_t stored = old;
                    cand = stored > base_off ? (uint32_t)(stored - base_off) : 0u;   // :70 saturating_sub
                    if (D < 64u) {
                        const uint32_t fD = __builtin_amdgcn_readlane(first, D);
                        const uint32_t c_first = __builtin_amdgcn_readlane(ck, fD & 63u);
                        if (lane == D) cand = c_first;
                    }
                }
                // candidate side, one round trip: 16 bytes at the candidate for the >= 4 test and the
                // forward extension, 8 bytes before both positions for the backtrack
                const bool reach = active && lane <= D && ck != init && cand <= ck && ck - cand <= 0xFFFFu;   // :200-201
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
                const uint32_t W = first_lane(__ballot(valid));
                // last lane whose `replace` really executed in sequential order (+1)
                uint32_t commit_end;   // lanes [0, commit_end) commit
                int outcome;           // 0 = continue, 1 = match at W, 2 = end of input
                if (W < e_end && W <= D) { commit_end = W + 1u; outcome = 1; }
                else if (D < 64u) { commit_end = D + 1u; outcome = 0; }
                else if (e_end < 64u) { commit_end = e_end; outcome = 2; }
                else { commit_end = bw; outcome = 0; }
                // EncoderTable contract (:67/:92): position + offset must fit the slot type
                if (!easy) {
                    const bool bad = active && lane < commit_end && ((uint64_t)ck + base_off > TT::kLimit);
                    if (__ballot(bad)) { status = LZF_CONTRACT; }
                }
                // ---- commit / roll back
                if (active) {
                    if (lane < commit_end) {
                        // lane first[D] is overridden by D when D commits
                        const uint32_t fD = D < 64u ? __builtin_amdgcn_readlane(first, D & 63u) : 64u;
                        const bool overridden = (D < commit_end) && (lane == fD) && (lane != D);
                        if (!overridden) tab[h] = ck + boff;
                    } else if (first >= commit_end) {
                        tab[h] = old;
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
                    break;
                }
                if (outcome == 2) { finished = true; break; }
                n += commit_end;
                if (n <= 66u) c += commit_end;
                else c += sched_prefix(n) - (sn ? sn : sched_prefix(n - commit_end));
            }
            if (status != LZF_OK) break;
            CPHASE(0);

            if (finished) {
                // ---- last literals, mod.rs:178-190
                const uint32_t L = len - ls;
                const uint32_t nl = lsic_len(L);
                const uint32_t total = 1u + nl + L;
                if (s.cap - s.pos < total) {
                    // which individual write fails does not matter: the sink content is dropped
                    status = LZF_OUTPUT_FULL;
                    break;
                }
                gu8* d = s.out + s.pos;
                if (lane == 0) d[0] = (uint8_t)((L < 15u ? L : 15u) << 4);
                if (nl) lsic_store(d + 1, L, nl, lane);
                wave_copy(d + 1u + nl, in + ls, L, lane);
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
                    const unsigned long long neq = __ballot(x != 0ull);
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
                    const unsigned long long neq = __ballot(ne);   // out-of-range lanes stop the scan
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
                    const unsigned long long neq = __ballot(ne);
                    if (neq) { bt += (uint32_t)__builtin_ctzll(neq); done = true; }
                    else bt += 64u;
                }
            }
            cursor = m_pos + m;                                            // :215
            CPHASE(1);
            // The literal run is loaded first and the next run's first probes are requested right behind it, so
            // both travel in one round trip (loads return in order: storing the literals then waits for the
            // literal loads only, and the probes land while this sequence is emitted).
            const uint32_t lit_len = (m_pos - bt) - ls;                    // = L below
            const bool lit_fast = lit_len <= 1024u;
            uint32_t lit_b = 0, lit_t = 0; u32x4 lit_v = {0, 0, 0, 0};
            if (lit_fast) {
                if (lit_len <= kWave) { if (lane < lit_len) lit_b = in[ls + lane]; }
                else {
                    const uint32_t bulk = lit_len & ~15u;
                    if (lane * 16u < bulk) lit_v = ld16(in + ls + lane * 16u);
                    if (lane < lit_len - bulk) lit_t = in[ls + bulk + lane];
                }
            }
            {
                const uint32_t ckn = cursor + lane;
                pfA0 = 0; pfA1 = 0;
                if ((uint64_t)cursor + kFirstBatch + 40u <= len) { if (lane < kFirstBatch) { pfA0 = ld8(in + ckn); pfA1 = ld8(in + ckn + 8u); } }
                else if (lane < kFirstBatch && ckn <= len && len - ckn >= 12u) { pfA0 = ld8(in + ckn); pfA1 = ld8_part(ckn + 8u); }
                pf_c = cursor;
            }
            // table.replace(input, cursor - 2) — unconditional (:218, quirks B1/B3)
            {
                const uint32_t q = cursor - 2u;
                if ((uint64_t)q + base_off > TT::kLimit) { status = LZF_CONTRACT; break; }
                uint64_t v8 = 0;
                const uint32_t need = KIND == LZF_TABLE_U32 ? 8u : 4u;
                if (KIND != LZF_TABLE_U32 || len - q >= 8u) {              // :43: fewer than 8 bytes left -> 0
                    if (m - 2u + need <= 16u) {                            // still inside the winner's 16 bytes
                        const uint32_t sh = (m - 2u) * 8u;
                        v8 = sh == 0u ? wA0 : sh < 64u ? ((wA0 >> sh) | (wA1 << (64u - sh))) : (wA1 >> (sh - 64u));
                        if (KIND != LZF_TABLE_U32) v8 &= 0xFFFFFFFFull;
                    } else {
                        v8 = KIND == LZF_TABLE_U32 ? ld8(in + q) : (uint64_t)ld4(in + q);
                    }
                }
                const uint32_t h = TT::hash(v8);
                if (lane == 0) tab[h] = q + boff;
            }
            const uint32_t dup_offset = m_pos - m_cand;                    // :208
            const uint32_t extra = m - 4u + bt;                            // :206,:214
            CPHASE(2);
            // ================= write_group, mod.rs:150-163 (+ :235 literal slice)
            const uint32_t lit_end = cursor - extra - 4u;
            const uint32_t L = lit_end - ls;
            const uint32_t nl = lsic_len(L), ne = lsic_len(extra);
            const uint32_t total = 1u + nl + L + 2u + ne;
            if (s.cap - s.pos < total) { status = LZF_OUTPUT_FULL; break; }
            gu8* d = s.out + s.pos;
            if (lane == 0) {
                d[0] = (uint8_t)(((L < 15u ? L : 15u) << 4) | (extra < 15u ? extra : 15u));
                d[1u + nl + L] = (uint8_t)dup_offset;
                d[2u + nl + L] = (uint8_t)(dup_offset >> 8);
            }
            if (nl) lsic_store(d + 1, L, nl, lane);
            if (lit_fast) {
                gu8* ld = d + 1u + nl;
                if (L <= kWave) { if (lane < L) ld[lane] = (uint8_t)lit_b; }
                else {
                    const uint32_t bulk = L & ~15u;
                    if (lane * 16u < bulk) st16(ld + lane * 16u, lit_v);
                    if (lane < L - bulk) ld[bulk + lane] = (uint8_t)lit_t;
                }
            } else {
                wave_copy(d + 1u + nl, in + ls, L, lane);
            }
            if (ne) lsic_store(d + 3u + nl + L, extra, ne, lane);
            s.pos += total;
            CPHASE(3);
        }
    }

    // ---- table out (`&mut table`): mutations survive OUTPUT_FULL, like the reference's
    if (job.table && !(job.flags & LZF_CJOB_TABLE_READONLY) && status != LZF_CONTRACT) {
        if (KIND == LZF_TABLE_U32) {
            LZF_GLOBAL lzf_u32_table* t = (LZF_GLOBAL lzf_u32_table*)job.table;
            for (uint32_t i = lane; i < TT::kSlots; i += kWave) t->dict[i] = tab[i];
        } else {
            LZF_GLOBAL lzf_u16_table* t = (LZF_GLOBAL lzf_u16_table*)job.table;
            for (uint32_t i = lane; i < TT::kSlots; i += kWave) t->dict[i] = (uint16_t)tab[i];
        }
    }
    if (lane == 0) {
        results[jid].out_len = s.pos;
        results[jid].status = status;
#ifdef LZF_PHASE_TIMING
        { uint32_t pk = 0; for (int i = 0; i < 4; ++i) { uint32_t u = (uint32_t)(g_tph[i] >> 23); if (u > 255u) u = 255u; pk |= u << (8 * i); } results[jid].reserved = pk; }
#else
        results[jid].reserved = (uint32_t)((clock64() - t_start) >> 10);   // diagnostic: shader kilo-cycles spent on this job
#endif
    }
}

template __global__ void lzf_compress_wave_kernel<LZF_TABLE_U32>(const lzf_compress_job*, lzf_job_result*, uint32_t, uint32_t, const uint32_t*);
template __global__ void lzf_compress_wave_kernel<LZF_TABLE_U16>(const lzf_compress_job*, lzf_job_result*, uint32_t, uint32_t, const uint32_t*);

}  // namespace lzf
