// lzf_simt.h — the handful of wave-level primitives the row-mapped compress kernel (lz4_compress_rows.inc) is written in.
//
// The kernel body is per-lane code with WAVE-UNIFORM control flow around every cross-lane operation: all 64 lanes call
// every primitive below, in the same order, with a per-lane predicate.  Two backends:
//
//   SimtGpu   (hipcc)  the primitives are the gfx950 instructions: v_cmp + s_and (ballot), ds_bpermute_b32, ds_read / ds_write /
//                      ds_min_u32 / ds_mskor_b32 on the workgroup's LDS.  LDS executes one wave's accesses in order, so the
//                      cross-lane dependencies through LDS need nothing else.  This is the product.
//   SimtEmu   (g++)    TEST INFRASTRUCTURE (tests/emu/): the same kernel source runs on the CPU, one fiber per lane, every
//                      primitive a lock-step point (the lanes are resumed round-robin, so when lane 0 continues behind its
//                      k-th primitive every lane has executed its k-th primitive).  The CPU suite compares the kernel's
//                      output with the oracle's without a GPU; nothing in the product library contains or loads it.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------------------------- gfx950
#include "lzf_device.h"
#define LZF_SIMT_FN __device__ __forceinline__
namespace lzf {
struct SimtGpu {
    uint32_t* lds;          // the workgroup's LDS words (a __shared__ array of the calling kernel)
    uint32_t lds_a;         // its LDS byte address
    LZF_SIMT_FN uint32_t lane() const { return threadIdx.x & 63u; }
    LZF_SIMT_FN unsigned long long ballot(bool p) const { return __ballot(p); }
    LZF_SIMT_FN bool any(bool p) const { return __ballot(p) != 0ull; }
    LZF_SIMT_FN uint32_t bperm(uint32_t src_lane, uint32_t v) const { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }
    LZF_SIMT_FN uint32_t lds_rd32(bool p, uint32_t w) const { uint32_t v = 0; if (p) v = lds[w]; return v; }
    LZF_SIMT_FN void lds_wr32(bool p, uint32_t w, uint32_t v) const { if (p) lds[w] = v; }
    LZF_SIMT_FN void lds_min32(bool p, uint32_t w, uint32_t v) const { if (p) atomicMin(&lds[w], v); }
    // mem = (mem & ~mask) | val
    LZF_SIMT_FN void lds_mskor32(bool p, uint32_t w, uint32_t mask, uint32_t val) const { if (p) lds_mskor32_raw(lds_a + 4u * w, mask, val); }
    LZF_SIMT_FN uint32_t atomic_inc(uint32_t* q) const { return atomicAdd(q, 1u); }
    LZF_SIMT_FN uint64_t clock() const { return (uint64_t)clock64(); }
    static LZF_SIMT_FN void lds_mskor32_raw(uint32_t a, uint32_t mask, uint32_t val) { asm volatile("ds_mskor_b32 %0, %1, %2" ::"v"(a), "v"(mask), "v"(val) : "memory"); }
};
LZF_SIMT_FN uint32_t simt_mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }      // both below 2^24: a full-rate multiply
LZF_SIMT_FN uint32_t simt_ctz32(uint32_t v) { return (uint32_t)__builtin_ctz(v); }
LZF_SIMT_FN uint32_t simt_ctz64(uint64_t v) { return (uint32_t)__builtin_ctzll(v); }
LZF_SIMT_FN uint32_t simt_clz64(uint64_t v) { return (uint32_t)__builtin_clzll(v); }
}  // namespace lzf

#else
// ------------------------------------------------------------------------------------------------------------- CPU emulation
#include <stdlib.h>
#include <string.h>
#include "../../include/lzfear_hip.h"
#define LZF_SIMT_FN static inline
#define LZF_GLOBAL
namespace lzf {
constexpr uint32_t kWave = 64;
typedef uint8_t gu8;
typedef const uint8_t cgu8;
struct u32x4 { uint32_t x, y, z, w; };
template <typename T> static inline cgu8* as_global(const T* p) { return (cgu8*)p; }
template <typename T> static inline gu8* as_global(T* p) { return (gu8*)p; }
static inline uint64_t ld8(cgu8* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline u32x4 ld16(cgu8* p) { u32x4 v; memcpy(&v, p, 16); return v; }
static inline void st16(gu8* p, u32x4 v) { memcpy(p, &v, 16); }
static inline uint32_t simt_mul24(uint32_t a, uint32_t b) { return a * b; }
static inline uint32_t simt_ctz32(uint32_t v) { return (uint32_t)__builtin_ctz(v); }
static inline uint32_t simt_ctz64(uint64_t v) { return (uint32_t)__builtin_ctzll(v); }
static inline uint32_t simt_clz64(uint64_t v) { return (uint32_t)__builtin_clzll(v); }

// One wavefront: 64 fibers resumed round-robin; sync() hands the CPU to the next lane.
struct EmuWave;
extern "C" void lzf_emu_switch(void** save_sp, void* load_sp);      // tests/emu/emu_fiber.S-free: defined with inline asm in emu_main.cpp
struct EmuWave {
    static constexpr uint32_t kLdsWords = 40960;                     // 160 KiB
    uint32_t lds[kLdsWords];
    uint64_t xchg[64];
    void* sp[64];                   // saved stack pointers of the lanes
    void* sched_sp;                 // the scheduler's
    bool finished[64];
    uint32_t cur;
    uint64_t n_sync;                // lock-step points executed (a cost figure for the tests)
};
struct SimtEmu {
    EmuWave* w;
    uint32_t my;
    uint32_t lane() const { return my; }
    void sync() const {             // every lane reaches the same primitive before any lane goes on
        w->n_sync++;
        const uint32_t nxt = (my + 1u) & 63u;
        w->cur = nxt;
        lzf_emu_switch(&w->sp[my], w->sp[nxt]);
    }
    unsigned long long ballot(bool p) const {
        sync(); w->xchg[my] = p ? 1u : 0u; sync();
        unsigned long long m = 0; for (uint32_t i = 0; i < 64; ++i) m |= (unsigned long long)(w->xchg[i] & 1u) << i;
        return m;
    }
    bool any(bool p) const { return ballot(p) != 0ull; }
    uint32_t bperm(uint32_t src_lane, uint32_t v) const { sync(); w->xchg[my] = v; sync(); return (uint32_t)w->xchg[src_lane & 63u]; }
    uint32_t lds_rd32(bool p, uint32_t i) const { sync(); return p ? w->lds[i] : 0u; }
    void lds_wr32(bool p, uint32_t i, uint32_t v) const { sync(); if (p) w->lds[i] = v; }
    void lds_min32(bool p, uint32_t i, uint32_t v) const { sync(); if (p && v < w->lds[i]) w->lds[i] = v; }
    void lds_mskor32(bool p, uint32_t i, uint32_t mask, uint32_t val) const { sync(); if (p) w->lds[i] = (w->lds[i] & ~mask) | val; }
    uint32_t atomic_inc(uint32_t* q) const { return (*q)++; }
    uint64_t clock() const { return 0; }
};
}  // namespace lzf
#endif
