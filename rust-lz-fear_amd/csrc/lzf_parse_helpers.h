// lzf_parse_helpers.h — hand-scheduled token-hop loops shared by the batched decompress kernels
// (see lz4_decompress_batched.hip for the description of the parse).
#pragma once
#include "lzf_device.h"

namespace lzf {
namespace {
// ---------------------------------------------------------------------------------------------
// The token-hop loop of the parse, hand-scheduled (direct variants: tokens are read from HBM/L2).
// The scalar unit is shared by the CU's four SIMDs and is the scarce issue resource of this kernel;
// hipcc keeps per-lane booleans as SGPR lane masks and spends ~22 SALU per hop on combining them.
// Here exec stays full, every predicate lives in VCC straight out of a v_cmp and is consumed by
// v_cndmask / v_addc, so a hop costs 2 SALU (the loop branches) and ~26 VALU.
//   live      <=> p < lim            (lim = the lane's stop position, 0 once the lane leaves the loop)
//   q          = position of the next token; forced to ~0 when the token needs more than this view
//                (a 0xFF length byte) so that the single test q < fast_end rejects it
//   a lane that cannot take its hop (end of region, end of input margin, 0xFF run) gets lim = 0 and
//   keeps p; the caller serves it with the general routine.
// decompress.rs:61-71 without the copies.
// ---------------------------------------------------------------------------------------------
#define LZF_HOP_HEAD(LD4, LD1, WAIT) \
    "Lhop_loop%=:\n\t" \
    "v_cmp_lt_u32 vcc, %[p], %[lim]\n\t" \
    "s_cbranch_vccz Lhop_done%=\n\t" \
    "v_min_u32 %[pa], %[pclamp], %[p]\n\t" \
    LD4 WAIT \
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
    LD1 \
    "v_addc_co_u32_e64 %[q], %[sx], 0, %[q], vcc\n\t" \
    WAIT \
    "v_cndmask_b32 %[m], 0, %[m], vcc\n\t" \
    "v_cmp_eq_u32 vcc, 0xff, %[m]\n\t" \
    "v_cndmask_b32_e64 %[q], %[q], -1, vcc\n\t" \
    "v_cmp_lt_u32 vcc, %[p], %[lim]\n\t" \
    "v_cndmask_b32 %[t], -1, %[q], vcc\n\t" \
    "v_cmp_gt_u32 vcc, %[fend], %[t]\n\t"            /* vcc = the lane takes this hop */
#define LZF_HOP_TAIL \
    "v_addc_co_u32_e64 %[n], %[sx], 0, %[n], vcc\n\t" \
    "v_cndmask_b32 %[p], %[p], %[q], vcc\n\t" \
    "v_cndmask_b32 %[lim], 0, %[lim], vcc\n\t" \
    "s_branch Lhop_loop%=\n" \
    "Lhop_done%=:"
#define LZF_HOP_GLB LZF_HOP_HEAD("global_load_dword %[w], %[pa], %[in]\n\t", "global_load_ubyte %[m], %[m], %[in]\n\t", "s_waitcnt vmcnt(0)\n\t")
#define LZF_HOP_RECORD \
    "v_cndmask_b32 %[kk], -1, %[k], vcc\n\t" \
    "v_addc_co_u32_e64 %[k], %[sx], 0, %[k], vcc\n\t" \
    "v_cmp_gt_u32_e64 %[sx], %[cap], %[kk]\n\t" \
    "v_subrev_u32 %[t], %[cstart], %[pa]\n\t" \
    "v_lshl_add_u32 %[m], %[kk], 1, %[toksa]\n\t" \
    "v_cndmask_b32_e64 %[m], %[dump], %[m], %[sx]\n\t" \
    "ds_write_b16 %[m], %[t]\n\t" \
    "v_cmp_eq_u32_e64 %[sx], %[cap], %[kk]\n\t" \
    "v_cndmask_b32_e64 %[cut], %[cut], %[pa], %[sx]\n\t"
__device__ __forceinline__ void hop_loop(uint32_t& p, uint32_t& lim, uint32_t& n, cgu8* in, uint32_t fast_end, uint32_t pclamp) {
    uint32_t pa, w, q, t, m; uint64_t sx;
    asm volatile(LZF_HOP_GLB LZF_HOP_TAIL
                 : [p] "+v"(p), [lim] "+v"(lim), [n] "+v"(n), [pa] "=&v"(pa), [w] "=&v"(w), [q] "=&v"(q), [t] "=&v"(t), [m] "=&v"(m), [sx] "=&s"(sx)
                 : [in] "s"(in), [fend] "s"(fast_end), [pclamp] "s"(pclamp)
                 : "vcc", "memory");
}
// Same, recording the token positions (relative to cstart) at toks[k++]; positions past the list's
// capacity go to the lane's dump slot and the position of token #cap is kept in `cut`.
__device__ __forceinline__ void hop_loop_record(uint32_t& p, uint32_t& lim, uint32_t& n, uint32_t& k, uint32_t& cut, cgu8* in,
                                                uint32_t fast_end, uint32_t pclamp, uint32_t cstart, uint32_t toks_a, uint32_t cap, uint32_t dump_a) {
    uint32_t pa, w, q, t, m, kk; uint64_t sx;
    asm volatile(LZF_HOP_GLB LZF_HOP_RECORD LZF_HOP_TAIL
                     : [p] "+v"(p), [lim] "+v"(lim), [n] "+v"(n), [k] "+v"(k), [cut] "+v"(cut), [pa] "=&v"(pa), [w] "=&v"(w), [q] "=&v"(q),
                       [t] "=&v"(t), [m] "=&v"(m), [kk] "=&v"(kk), [sx] "=&s"(sx)
                     : [in] "s"(in), [fend] "s"(fast_end), [pclamp] "s"(pclamp), [cstart] "s"(cstart), [toksa] "s"(toks_a), [cap] "s"(cap), [dump] "v"(dump_a)
                     : "vcc", "memory");
}

// Staged variants precompute, for every byte position of the chunk, the distance to the next token
// (nxt[], one byte per position, 255 = "not a plain token here": 0xFF length bytes, end of input, ...),
// with all lanes busy and no dependent chain.  A hop of the walks is then one LDS byte read.
// p, lim and pclamp are LDS byte addresses into nxt[] (position - cstart + address of nxt).
#define LZF_THOP_HEAD \
    "Lthop_loop%=:\n\t" \
    "v_cmp_lt_u32 vcc, %[p], %[lim]\n\t" \
    "s_cbranch_vccz Lthop_done%=\n\t" \
    "v_min_u32 %[pa], %[pclamp], %[p]\n\t" \
    "ds_read_u8 %[d], %[pa]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_cndmask_b32 %[t], -1, %[d], vcc\n\t"          /* idle lanes: ~0 */ \
    "v_cmp_gt_u32 vcc, 0xff, %[t]\n\t"               /* vcc = the lane takes this hop (live and not 255) */
#define LZF_THOP_TAIL \
    "v_addc_co_u32_e64 %[n], %[sx], 0, %[n], vcc\n\t" \
    "v_cndmask_b32 %[t], 0, %[d], vcc\n\t" \
    "v_add_u32 %[p], %[p], %[t]\n\t" \
    "v_cndmask_b32 %[lim], 0, %[lim], vcc\n\t" \
    "s_branch Lthop_loop%=\n" \
    "Lthop_done%=:"
__device__ __forceinline__ void thop_loop(uint32_t& p, uint32_t& lim, uint32_t& n, uint32_t pclamp) {
    uint32_t pa, d, t; uint64_t sx;
    asm volatile(LZF_THOP_HEAD LZF_THOP_TAIL
                 : [p] "+v"(p), [lim] "+v"(lim), [n] "+v"(n), [pa] "=&v"(pa), [d] "=&v"(d), [t] "=&v"(t), [sx] "=&s"(sx)
                 : [pclamp] "s"(pclamp)
                 : "vcc", "memory");
}
// Same, recording token positions (chunk offsets) at toks[k++]; `cut` keeps the nxt[] address of token #cap.
__device__ __forceinline__ void thop_loop_record(uint32_t& p, uint32_t& lim, uint32_t& n, uint32_t& k, uint32_t& cut, uint32_t pclamp,
                                                 uint32_t nxt_a, uint32_t toks_a, uint32_t cap, uint32_t dump_a) {
    uint32_t pa, d, t, m, kk; uint64_t sx;
    asm volatile(LZF_THOP_HEAD
                 "v_cndmask_b32 %[kk], -1, %[k], vcc\n\t"
                 "v_addc_co_u32_e64 %[k], %[sx], 0, %[k], vcc\n\t"
                 "v_cmp_gt_u32_e64 %[sx], %[cap], %[kk]\n\t"
                 "v_subrev_u32 %[t], %[nxta], %[pa]\n\t"
                 "v_lshl_add_u32 %[m], %[kk], 1, %[toksa]\n\t"
                 "v_cndmask_b32_e64 %[m], %[dump], %[m], %[sx]\n\t"
                 "ds_write_b16 %[m], %[t]\n\t"
                 "v_cmp_eq_u32_e64 %[sx], %[cap], %[kk]\n\t"
                 "v_cndmask_b32_e64 %[cut], %[cut], %[pa], %[sx]\n\t"
                 LZF_THOP_TAIL
                 : [p] "+v"(p), [lim] "+v"(lim), [n] "+v"(n), [k] "+v"(k), [cut] "+v"(cut), [pa] "=&v"(pa), [d] "=&v"(d), [t] "=&v"(t),
                   [m] "=&v"(m), [kk] "=&v"(kk), [sx] "=&s"(sx)
                 : [pclamp] "s"(pclamp), [nxta] "s"(nxt_a), [toksa] "s"(toks_a), [cap] "s"(cap), [dump] "v"(dump_a)
                 : "vcc", "memory");
}
// 32-bit token-list entries (chunk offset in the low half; the paired kernel's parser fills in lengths later)
__device__ __forceinline__ void thop_loop_record32(uint32_t& p, uint32_t& lim, uint32_t& n, uint32_t& k, uint32_t& cut, uint32_t pclamp,
                                                 uint32_t nxt_a, uint32_t toks_a, uint32_t cap, uint32_t dump_a) {
    uint32_t pa, d, t, m, kk; uint64_t sx;
    asm volatile(LZF_THOP_HEAD
                 "v_cndmask_b32 %[kk], -1, %[k], vcc\n\t"
                 "v_addc_co_u32_e64 %[k], %[sx], 0, %[k], vcc\n\t"
                 "v_cmp_gt_u32_e64 %[sx], %[cap], %[kk]\n\t"
                 "v_subrev_u32 %[t], %[nxta], %[pa]\n\t"
                 "v_lshl_add_u32 %[m], %[kk], 2, %[toksa]\n\t"
                 "v_cndmask_b32_e64 %[m], %[dump], %[m], %[sx]\n\t"
                 "ds_write_b32 %[m], %[t]\n\t"
                 "v_cmp_eq_u32_e64 %[sx], %[cap], %[kk]\n\t"
                 "v_cndmask_b32_e64 %[cut], %[cut], %[pa], %[sx]\n\t"
                 LZF_THOP_TAIL
                 : [p] "+v"(p), [lim] "+v"(lim), [n] "+v"(n), [k] "+v"(k), [cut] "+v"(cut), [pa] "=&v"(pa), [d] "=&v"(d), [t] "=&v"(t),
                   [m] "=&v"(m), [kk] "=&v"(kk), [sx] "=&s"(sx)
                 : [pclamp] "s"(pclamp), [nxta] "s"(nxt_a), [toksa] "s"(toks_a), [cap] "s"(cap), [dump] "v"(dump_a)
                 : "vcc", "memory");
}
__device__ __forceinline__ void lds_ld8x4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t& v0, uint32_t& v1, uint32_t& v2, uint32_t& v3) {
    asm volatile("ds_read_u8 %0, %4\n\tds_read_u8 %1, %5\n\tds_read_u8 %2, %6\n\tds_read_u8 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}


}  // namespace
}  // namespace lzf
