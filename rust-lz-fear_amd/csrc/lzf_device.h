// lzf_device.h — shared device-side helpers for the gfx950 LZ4 kernels (wave64 only).
#pragma once
#if !defined(LZF_ANALYSIS) && (defined(LZF_DBG_SKIP) || defined(LZF_DBG_SKIP2) || defined(LZF_DBG_TIME) || defined(LZF_DBG_COUNT) || defined(LZF_DBG_DRY_MAIN) || defined(LZF_DBG_LDS_PAD) || defined(LZF_DBG_PATHS) || defined(LZF_DBG_PHASE_SEL) || defined(LZF_DBG_ROUNDS) || defined(LZF_DBG_TIMELINE) || defined(LZF_DBG_NOFENCE) || \
    defined(LZF_SEG_DBG_SKIP) || defined(LZF_SEG_DBG_NOWAIT) || defined(LZF_SEG_TIME) || defined(LZF_SEG_NOASM) || defined(LZF_SEG_NORLE))
#error "LZF_DBG_* / LZF_SEG_* instrumentation is for -DLZF_ANALYSIS builds only"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lzfear_hip.h"

namespace lzf {

constexpr uint32_t kWave = 64;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// Byte-addressed 16/8/4/2-byte accesses.  gfx950 runs with unaligned global access enabled, so
// these lower to single global_load/store_dwordx4/x2/dword/ushort at any byte address.
struct __attribute__((packed, aligned(1))) U16B { u32x4 v; };
struct __attribute__((packed, aligned(1))) U8B { uint64_t v; };
struct __attribute__((packed, aligned(1))) U4B { uint32_t v; };
struct __attribute__((packed, aligned(1))) U2B { uint16_t v; };

// Job pointers arrive inside a struct, so hipcc only knows them as generic pointers and would emit
// FLAT loads/stores (which count against both vmcnt and lgkmcnt and are slower).  Everything a
// job points at lives in HBM: the kernels cast to the global address space once, and every helper
// below takes global pointers, so the accesses compile to global_load / global_store.
#define LZF_GLOBAL __attribute__((address_space(1)))
typedef LZF_GLOBAL uint8_t gu8;
typedef const LZF_GLOBAL uint8_t cgu8;
template <typename T> __device__ __forceinline__ cgu8* as_global(const T* p) { return (cgu8*)p; }
template <typename T> __device__ __forceinline__ gu8* as_global(T* p) { return (gu8*)p; }

__device__ __forceinline__ u32x4 ld16(cgu8* p) { return reinterpret_cast<const LZF_GLOBAL U16B*>(p)->v; }
__device__ __forceinline__ void st16(gu8* p, u32x4 v) { reinterpret_cast<LZF_GLOBAL U16B*>(p)->v = v; }
__device__ __forceinline__ uint64_t ld8(cgu8* p) { return reinterpret_cast<const LZF_GLOBAL U8B*>(p)->v; }
__device__ __forceinline__ uint32_t ld4(cgu8* p) { return reinterpret_cast<const LZF_GLOBAL U4B*>(p)->v; }
__device__ __forceinline__ uint32_t ld2(cgu8* p) { return reinterpret_cast<const LZF_GLOBAL U2B*>(p)->v; }

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Index of the lowest set bit of a wave ballot, or 64 when empty.
__device__ __forceinline__ uint32_t first_lane(unsigned long long m) {
    return m ? (uint32_t)__builtin_ctzll(m) : 64u;
}

// Every store this wavefront has issued so far becomes visible to its own later loads
// (s_waitcnt vmcnt(0); the per-CU L1 is write-through, so same-CU readers see it).
__device__ __forceinline__ void wave_store_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
}

// Cooperative forward copy of n bytes, source and destination not overlapping *within the
// copied range* (the source is fully written before the call).  All 64 lanes participate.
__device__ __forceinline__ void wave_copy(gu8* __restrict__ dst, cgu8* __restrict__ src,
                                          uint32_t n, uint32_t lane) {
    if (n <= kWave) {
        if (lane < n) dst[lane] = src[lane];
        return;
    }
    const uint32_t bulk = n & ~15u;
    for (uint32_t i = lane * 16u; i < bulk; i += kWave * 16u) st16(dst + i, ld16(src + i));
    const uint32_t tail = n - bulk;
    if (lane < tail) dst[bulk + lane] = src[bulk + lane];
}

}  // namespace lzf

// ---------------------------------------------------------------------------------------------
// wave64 scans on DPP (row_shr 1/2/4/8, row_bcast 15/31): no LDS traffic, ~6 VALU each
// ---------------------------------------------------------------------------------------------
namespace lzf {
#define LZF_DPP(old, x, ctrl, rmask) (uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(x), ctrl, rmask, 0xf, false)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {     // inclusive prefix sum over lanes
    v += LZF_DPP(0, v, 0x111, 0xf); v += LZF_DPP(0, v, 0x112, 0xf);
    v += LZF_DPP(0, v, 0x114, 0xf); v += LZF_DPP(0, v, 0x118, 0xf);
    v += LZF_DPP(0, v, 0x142, 0xa); v += LZF_DPP(0, v, 0x143, 0xc);
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) {     // inclusive prefix max over lanes
    uint32_t t;
    t = LZF_DPP(0, v, 0x111, 0xf); v = t > v ? t : v; t = LZF_DPP(0, v, 0x112, 0xf); v = t > v ? t : v;
    t = LZF_DPP(0, v, 0x114, 0xf); v = t > v ? t : v; t = LZF_DPP(0, v, 0x118, 0xf); v = t > v ? t : v;
    t = LZF_DPP(0, v, 0x142, 0xa); v = t > v ? t : v; t = LZF_DPP(0, v, 0x143, 0xc); v = t > v ? t : v;
    return v;
}
// value of the previous lane (lane 0 gets `first`)
__device__ __forceinline__ uint32_t wave_prev(uint32_t v, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// ---------------------------------------------------------------------------------------------
// Byte-addressed LDS accesses of 2/4/8 bytes at ANY alignment.  gfx950 executes misaligned DS
// accesses correctly (about 2x the cost of an aligned one; tools/lds_unaligned_test.hip), but
// hipcc never emits them for under-aligned types (it splits into bytes), hence inline asm.
// `a` is the LDS byte address (low 32 bits of a __shared__ pointer).  The load forms wait for
// their own data (s_waitcnt lgkmcnt(0)) because hipcc does not count asm memory operations.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)reinterpret_cast<uintptr_t>(p); }
// mem = (mem & ~mask) | val in one LDS instruction (no branch on "set or clear", no return value to wait for; LDS executes a
// wave's accesses in order, so later reads see it)
__device__ __forceinline__ void lds_mskor32(uint32_t a, uint32_t mask, uint32_t val) { asm volatile("ds_mskor_b32 %0, %1, %2" ::"v"(a), "v"(mask), "v"(val) : "memory"); }
__device__ __forceinline__ void lds_st8(uint32_t a, uint32_t v) { asm volatile("ds_write_b8 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st16(uint32_t a, uint32_t v) { asm volatile("ds_write_b16 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st32(uint32_t a, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st64(uint32_t a, uint64_t v) { asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_ld64x4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3,
                                           uint64_t& v0, uint64_t& v1, uint64_t& v2, uint64_t& v3) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}
__device__ __forceinline__ void lds_ld32x2(uint32_t a0, uint32_t a1, uint32_t& v0, uint32_t& v1) {
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ void lds_ld16x2(uint32_t a0, uint32_t a1, uint32_t& v0, uint32_t& v1) {
    asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ uint32_t lds_ld16(uint32_t a) {
    uint32_t v; asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); return v;
}
__device__ __forceinline__ uint32_t lds_ld8(uint32_t a) {
    uint32_t v; asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); return v;
}
}  // namespace lzf
