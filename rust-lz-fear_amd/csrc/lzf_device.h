// lzf_device.h — shared device-side helpers for the gfx950 LZ4 kernels (wave64 only).
#pragma once
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

__device__ __forceinline__ u32x4 ld16(const uint8_t* p) { return reinterpret_cast<const U16B*>(p)->v; }
__device__ __forceinline__ void st16(uint8_t* p, u32x4 v) { reinterpret_cast<U16B*>(p)->v = v; }
__device__ __forceinline__ uint64_t ld8(const uint8_t* p) { return reinterpret_cast<const U8B*>(p)->v; }
__device__ __forceinline__ uint32_t ld4(const uint8_t* p) { return reinterpret_cast<const U4B*>(p)->v; }

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
__device__ __forceinline__ void wave_copy(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src,
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
