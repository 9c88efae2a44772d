// lzf_compress_common.h — pieces shared by the compress kernels (skip schedule, bounded sink, LSIC coding).
#pragma once
#include "lzf_simt.h"      // (lzf_device.h under hipcc; plain C++ stand-ins for the CPU emulation of the test suite)

namespace lzf {

// S(m) = sum of the first m advances of one literal run (mod.rs:174-175,225-231):
// advance after probe j is 1 for j <= 65, then (62 + j) >> 6.
LZF_SIMT_FN uint32_t sched_prefix(uint32_t m) {
    if (m <= 66u) return m;
    const uint32_t r = m - 66u;
    const uint32_t full = r >> 6, rem = r & 63u;
    return 66u + 64u * (full * (full - 1u) / 2u + 2u * full) + rem * (full + 2u);
}

constexpr uint32_t kMark = 0xFFFFFFFFu;
constexpr uint32_t kMaxLen = 0x7FFFFF00u;
#ifndef LZF_FIRST_BATCH
#define LZF_FIRST_BATCH 16
#endif
constexpr uint32_t kFirstBatch = LZF_FIRST_BATCH;     // lanes probing in the first batch of a literal run

// Bounded sink with NoPartialWrites semantics (src/framed/compress.rs:294-314).
struct Sink {
    gu8* out;
    uint32_t pos, cap;
};

// LSIC tail length in bytes (mod.rs:243-260): 0 if v < 15 else (v-15)/255 + 1.
LZF_SIMT_FN uint32_t lsic_len(uint32_t v) { return v < 15u ? 0u : (v - 15u) / 255u + 1u; }

LZF_SIMT_FN void lsic_store(gu8* dst, uint32_t v, uint32_t n, uint32_t lane) {
    // n = lsic_len(v) > 0: n-1 bytes of 0xFF then (v-15) % 255
    for (uint32_t i = lane; i < n; i += kWave) dst[i] = (i + 1u == n) ? (uint8_t)((v - 15u) % 255u) : (uint8_t)0xFF;
}


// Jobs the compact-table kernel (lz4_compress_compact.hip) takes instead of the general one: U32Table semantics with a
// fresh table or a read-only template whose `offset` is 0 (every block the frame layer compresses in independent-
// blocks mode, src/framed/compress.rs:220,265-270), positions below 2 GiB.
LZF_SIMT_FN bool compress_job_is_compact(const lzf_compress_job& job) {
    if (job.table_kind != LZF_TABLE_U32 || job.input_len >= kMaxLen || job.cursor > job.input_len) return false;
    if (!job.table) return true;
    if (!(job.flags & LZF_CJOB_TABLE_READONLY)) return false;
    return ((const LZF_GLOBAL lzf_u32_table*)job.table)->offset == 0ull;
}

// What lzf_compress_team_kernel (lz4_compress_team.inc: the latency class) takes when another kernel stands behind it: every
// compact job, and — round 6 — U32 jobs with a caller-owned table at any `offset` (mod.rs:30,:65-74): linked-block streams
// (framed/compress.rs:271-275), read at start, written back at the end.  The kernel keeps positions in its LDS table and
// applies the offset on the way in and out, which is exact as long as position + offset fits the slot (:67: else the reference
// panics — those jobs stay with the general kernel, which says LZF_CONTRACT) and position 0 is never inserted behind an
// offset (cursor > 0: see the write-back in lz4_compress_team.inc).
LZF_SIMT_FN bool compress_job_is_team(const lzf_compress_job& job) {
    if (job.table_kind != LZF_TABLE_U32 || job.input_len >= kMaxLen || job.cursor > job.input_len) return false;
    if (!job.table) return true;
    const uint64_t off = ((const LZF_GLOBAL lzf_u32_table*)job.table)->offset;
    if (off == 0ull) return true;
    return job.cursor > 0u && off + job.input_len <= 0xFFFFFFFFull;
}
// Internal status of a team job whose writer refused while its table is the caller's: the searcher runs ahead of the emitter,
// so the table in LDS is past the refused sequence — nothing is written back, and the general kernel launched behind the
// team kernel does the job again from the caller's table (it leaves the table as the reference does, mod.rs:150-163 with
// framed/compress.rs:294-314).  Never seen by a caller.
constexpr int kTeamRetry = -30000;

}  // namespace lzf
