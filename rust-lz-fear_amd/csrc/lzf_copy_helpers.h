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
// Non-overlapping match of n = 4..64 bytes, ring -> ring, neither range wrapping: 4..7 two 4-byte pieces; 8..32 four two-ended
// 8-byte pieces (duplicates below 17 bytes); beyond 32 the first 32 bytes here and the last 32 by put_tail_lds (caller).
__device__ __forceinline__ void put_match_lds_1rt(uint32_t d, uint32_t s, uint32_t n, unsigned long long who) {
    uint32_t a1, a2, a3, dl, w0, w1; uint64_t v0, v1, v2, v3; unsigned long long sv, mA, mB;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_and_b64 exec, exec, %[who]\n\t"
        "v_sub_u32 %[dl], %[d], %[s]\n\t"
        "v_cmp_gt_u32 vcc, 8, %[n]\n\t"                       /* n < 8 */
        "s_and_b64 %[mB], exec, vcc\n\t"
        "s_andn2_b64 %[mA], exec, vcc\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "v_add3_u32 %[a3], %[s], %[n], -4\n\t"
        "ds_read_b32 %[w0], %[s]\n\t"
        "ds_read_b32 %[w1], %[a3]\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        "v_cmp_lt_u32 vcc, 16, %[n]\n\t"                      /* more than two pieces */
        "v_cndmask_b32_e64 %[a1], 0, 8, vcc\n\t"
        "v_add_u32 %[a1], %[a1], %[s]\n\t"
        "v_min_u32 %[a2], 32, %[n]\n\t"
        "v_add3_u32 %[a3], %[s], %[a2], -8\n\t"               /* s + min(n, 32) - 8 */
        "v_add_u32 %[a2], -8, %[a3]\n\t"
        "v_cndmask_b32 %[a2], %[s], %[a2], vcc\n\t"
        "ds_read_b64 %[v0], %[s]\n\t"
        "ds_read_b64 %[v1], %[a1]\n\t"
        "ds_read_b64 %[v2], %[a2]\n\t"
        "ds_read_b64 %[v3], %[a3]\n\t"
        "v_add_u32 %[a1], %[a1], %[dl]\n\t"
        "v_add_u32 %[a2], %[a2], %[dl]\n\t"
        "v_add_u32 %[a3], %[a3], %[dl]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "ds_write_b64 %[d], %[v0]\n\t"
        "ds_write_b64 %[a1], %[v1]\n\t"
        "ds_write_b64 %[a2], %[v2]\n\t"
        "ds_write_b64 %[a3], %[v3]\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "v_add_u32 %[a3], %[a3], %[dl]\n\t"
        "ds_write_b32 %[d], %[w0]\n\t"
        "ds_write_b32 %[a3], %[w1]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [dl] "=&v"(dl), [w0] "=&v"(w0), [w1] "=&v"(w1), [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3),
          [sv] "=&s"(sv), [mA] "=&s"(mA), [mB] "=&s"(mB)
        : [d] "v"(d), [s] "v"(s), [n] "v"(n), [who] "s"(who)
        : "vcc", "memory");
}
// Literal run of n = 1..64 bytes, staged chunk -> ring, neither range wrapping (the source may be read up to 7 bytes beyond its
// end: still LDS): 8 and more as above; 1..7 from the 8 bytes at the source, written as 4 + 2 + 1 bytes by the bits of n.
__device__ __forceinline__ void put_small_lds_1rt(uint32_t d, uint32_t s, uint32_t n, unsigned long long who) {
    uint32_t a1, a2, a3, dl, dd, wl, wh; uint64_t v0, v1, v2, v3; unsigned long long sv, mA, mB, mT;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_and_b64 exec, exec, %[who]\n\t"
        "v_sub_u32 %[dl], %[d], %[s]\n\t"
        "v_cmp_gt_u32 vcc, 8, %[n]\n\t"                       /* n < 8 */
        "s_and_b64 %[mB], exec, vcc\n\t"
        "s_andn2_b64 %[mA], exec, vcc\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "ds_read_b32 %[wl], %[s]\n\t"
        "ds_read_b32 %[wh], %[s] offset:4\n\t"
        "v_mov_b32 %[dd], %[d]\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        "v_cmp_lt_u32 vcc, 16, %[n]\n\t"
        "v_cndmask_b32_e64 %[a1], 0, 8, vcc\n\t"
        "v_add_u32 %[a1], %[a1], %[s]\n\t"
        "v_min_u32 %[a2], 32, %[n]\n\t"
        "v_add3_u32 %[a3], %[s], %[a2], -8\n\t"
        "v_add_u32 %[a2], -8, %[a3]\n\t"
        "v_cndmask_b32 %[a2], %[s], %[a2], vcc\n\t"
        "ds_read_b64 %[v0], %[s]\n\t"
        "ds_read_b64 %[v1], %[a1]\n\t"
        "ds_read_b64 %[v2], %[a2]\n\t"
        "ds_read_b64 %[v3], %[a3]\n\t"
        "v_add_u32 %[a1], %[a1], %[dl]\n\t"
        "v_add_u32 %[a2], %[a2], %[dl]\n\t"
        "v_add_u32 %[a3], %[a3], %[dl]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "ds_write_b64 %[d], %[v0]\n\t"
        "ds_write_b64 %[a1], %[v1]\n\t"
        "ds_write_b64 %[a2], %[v2]\n\t"
        "ds_write_b64 %[a3], %[v3]\n\t"
        /* 1..7 bytes: 4, then 2, then 1 */
        "s_mov_b64 exec, %[mB]\n\t"
        "v_and_b32 %[a1], 4, %[n]\n\t"
        "v_cmp_ne_u32 vcc, 0, %[a1]\n\t"
        "s_and_b64 %[mT], exec, vcc\n\t"
        "s_mov_b64 exec, %[mT]\n\t"
        "ds_write_b32 %[dd], %[wl]\n\t"
        "v_mov_b32 %[wl], %[wh]\n\t"
        "v_add_u32 %[dd], 4, %[dd]\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "v_and_b32 %[a1], 2, %[n]\n\t"
        "v_cmp_ne_u32 vcc, 0, %[a1]\n\t"
        "s_and_b64 %[mT], exec, vcc\n\t"
        "s_mov_b64 exec, %[mT]\n\t"
        "ds_write_b16 %[dd], %[wl]\n\t"
        "v_lshrrev_b32 %[wl], 16, %[wl]\n\t"
        "v_add_u32 %[dd], 2, %[dd]\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "v_and_b32 %[a1], 1, %[n]\n\t"
        "v_cmp_ne_u32 vcc, 0, %[a1]\n\t"
        "s_and_b64 %[mT], exec, vcc\n\t"
        "s_mov_b64 exec, %[mT]\n\t"
        "ds_write_b8 %[dd], %[wl]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [dl] "=&v"(dl), [dd] "=&v"(dd), [wl] "=&v"(wl), [wh] "=&v"(wh), [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3),
          [sv] "=&s"(sv), [mA] "=&s"(mA), [mB] "=&s"(mB), [mT] "=&s"(mT)
        : [d] "v"(d), [s] "v"(s), [n] "v"(n), [who] "s"(who)
        : "vcc", "memory");
}

struct No { static constexpr bool value = false; };
struct Yes { static constexpr bool value = true; };
}  // namespace
}  // namespace lzf
