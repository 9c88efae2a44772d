// lzf_copy_helpers.h — constants and per-lane copy helpers shared by the batched decompress kernels.
#pragma once
#include "lzf_device.h"

namespace lzf {
namespace {

constexpr uint32_t kMaxPosB = 0x7FFFFF00u;
constexpr uint32_t kShort = 64;          // bytes a lane moves by itself (literal runs, near matches); longer runs are cooperative
constexpr uint32_t kFarShort = 32;       // same for far matches (their bytes wait in registers while the literals are copied)
constexpr uint32_t kTotClamp = 1u << 25; // per-sequence output clamp inside the position scan
#ifndef LZF_DBG_SKIP
#define LZF_DBG_SKIP 0      // analysis builds only: bit0 batches, bit1 serial matches, bit2 far, bit3 literals, bit4 flush, bit5 round 1
#endif

// Bytes 32..n-1 of a 33..64-byte run whose first 32 bytes are moved separately: last four 8-byte pieces (they may
// overlap the first 32 bytes: same data).
__device__ __forceinline__ void put_tail_lds(uint32_t dst, uint32_t srca, uint32_t n) {
    uint64_t v0, v1, v2, v3;
    lds_ld64x4(srca + n - 32u, srca + n - 24u, srca + n - 16u, srca + n - 8u, v0, v1, v2, v3);
    lds_st64(dst + n - 32u, v0); lds_st64(dst + n - 24u, v1); lds_st64(dst + n - 16u, v2); lds_st64(dst + n - 8u, v3);
}
// Exact per-lane copy of n (1..64) bytes between two non-overlapping LDS byte ranges, neither of
// which wraps: two-ended pieces (first/last 8, 4 or 2 bytes), at most 4 reads + 4 writes up to 32 bytes, 8 + 8 beyond.
__device__ __forceinline__ void put_small_lds(uint32_t dst, uint32_t srca, uint32_t n) {
    if (n > 32u) {
        uint64_t v0, v1, v2, v3;
        lds_ld64x4(srca, srca + 8u, srca + 16u, srca + 24u, v0, v1, v2, v3);
        lds_st64(dst, v0); lds_st64(dst + 8u, v1); lds_st64(dst + 16u, v2); lds_st64(dst + 24u, v3);
        put_tail_lds(dst, srca, n);
    } else if (n >= 8u) {
        const bool big = n > 16u;
        uint64_t v0, v1, v2, v3;
        lds_ld64x4(srca, big ? srca + 8u : srca, big ? srca + n - 16u : srca, srca + n - 8u, v0, v1, v2, v3);
        lds_st64(dst, v0);
        if (big) { lds_st64(dst + 8u, v1); lds_st64(dst + n - 16u, v2); }
        lds_st64(dst + n - 8u, v3);
    } else if (n >= 4u) {
        uint32_t v0, v1; lds_ld32x2(srca, srca + n - 4u, v0, v1);
        lds_st32(dst, v0); lds_st32(dst + n - 4u, v1);
    } else if (n >= 2u) {
        uint32_t v0, v1; lds_ld16x2(srca, srca + n - 2u, v0, v1);
        lds_st16(dst, v0); lds_st16(dst + n - 2u, v1);
    } else if (n == 1u) {
        lds_st8(dst, lds_ld8(srca));
    }
}
// A match is 4..64 bytes here: three classes.
__device__ __forceinline__ void put_match_lds(uint32_t dst, uint32_t srca, uint32_t n) {
    if (n > 32u) {
        // (the rare class: its sixteen addresses are worked out HERE — the empty asm keeps hipcc from hoisting them out of the caller's
        //  round loop, where they occupied a dozen registers for every batch whether or not it had such a match)
        asm volatile("" : "+v"(dst), "+v"(srca), "+v"(n));
        uint64_t v0, v1, v2, v3;
        lds_ld64x4(srca, srca + 8u, srca + 16u, srca + 24u, v0, v1, v2, v3);
        lds_st64(dst, v0); lds_st64(dst + 8u, v1); lds_st64(dst + 16u, v2); lds_st64(dst + 24u, v3);
        put_tail_lds(dst, srca, n);
    } else if (n >= 8u) {
        const bool big = n > 16u;
        uint64_t v0, v1, v2, v3;
        lds_ld64x4(srca, big ? srca + 8u : srca, big ? srca + n - 16u : srca, srca + n - 8u, v0, v1, v2, v3);
        lds_st64(dst, v0);
        if (big) { lds_st64(dst + 8u, v1); lds_st64(dst + n - 16u, v2); }
        lds_st64(dst + n - 8u, v3);
    } else {
        uint32_t v0, v1; lds_ld32x2(srca, srca + n - 4u, v0, v1);
        lds_st32(dst, v0); lds_st32(dst + n - 4u, v1);
    }
}
// Same, source in global memory (unaligned loads; reads exactly [g, g+n)).
__device__ __forceinline__ void put_small_glb(uint32_t dst, cgu8* g, uint32_t n) {
    if (n > 32u) {
        // (two halves one after the other: eight 8-byte values in flight at once were the register peak of the copy stage, and
        //  the case is rare — a literal run or far match of 33..64 bytes that is not in the staged chunk)
        {
            const uint64_t a0 = ld8(g), a1 = ld8(g + 8u), a2 = ld8(g + 16u), a3 = ld8(g + 24u);
            lds_st64(dst, a0); lds_st64(dst + 8u, a1); lds_st64(dst + 16u, a2); lds_st64(dst + 24u, a3);
        }
        asm volatile("" ::: "memory");
        {
            const uint64_t b0 = ld8(g + n - 32u), b1 = ld8(g + n - 24u), b2 = ld8(g + n - 16u), b3 = ld8(g + n - 8u);
            lds_st64(dst + n - 32u, b0); lds_st64(dst + n - 24u, b1); lds_st64(dst + n - 16u, b2); lds_st64(dst + n - 8u, b3);
        }
    } else if (n >= 8u) {
        const bool big = n > 16u;
        const uint64_t v0 = ld8(g), v3 = ld8(g + n - 8u);
        uint64_t v1 = 0, v2 = 0;
        if (big) { v1 = ld8(g + 8u); v2 = ld8(g + n - 16u); }
        lds_st64(dst, v0);
        if (big) { lds_st64(dst + 8u, v1); lds_st64(dst + n - 16u, v2); }
        lds_st64(dst + n - 8u, v3);
    } else if (n >= 4u) {
        const uint32_t v0 = ld4(g), v1 = ld4(g + n - 4u);
        lds_st32(dst, v0); lds_st32(dst + n - 4u, v1);
    } else if (n >= 2u) {
        const uint32_t v0 = ld2(g), v1 = ld2(g + n - 2u);
        lds_st16(dst, v0); lds_st16(dst + n - 2u, v1);
    } else if (n == 1u) {
        lds_st8(dst, g[0]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One LDS round trip for every size class at once (round 6).  The helpers above serve one class per branch, and every branch
// waits for its own reads: a batch whose lanes fall into three classes pays three dependent round trips per call.  A wave
// that shares its SIMD with five others is bound by exactly those waits (section timers, profiles/r06_fed_sections.txt), so
// the forms below issue the reads of ALL classes (EXEC narrowed per class, no wait in between), wait once, then issue all
// writes.  `who` = the lanes that take part (a subset of EXEC, uniform value); every lane of the wave executes the call.
// ---------------------------------------------------------------------------------------------------------------------
struct No { static constexpr bool value = false; };
struct Yes { static constexpr bool value = true; };
}  // namespace
}  // namespace lzf
