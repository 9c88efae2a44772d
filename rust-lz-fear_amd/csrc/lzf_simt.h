// lzf_simt.h — the handful of wave-level primitives the team compress kernel (lz4_compress_team.inc: searcher / emitter / feeder) is written in.
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
    // the same without a predicate: a lane with nothing to do aims at a scratch word of its own instead of leaving the instruction
    // (no exec-mask bookkeeping around the access)
    LZF_SIMT_FN uint32_t lds_rd32u(uint32_t w) const { return lds[w]; }
    LZF_SIMT_FN void lds_wr32u(uint32_t w, uint32_t v) const { lds[w] = v; }
    LZF_SIMT_FN void lds_min32u(uint32_t w, uint32_t v) const { atomicMin(&lds[w], v); }
    LZF_SIMT_FN void lds_mskor32u(uint32_t w, uint32_t mask, uint32_t val) const { lds_mskor32_raw(lds_a + 4u * w, mask, val); }
    // 16 aligned bytes to LDS words w .. w + 3
    LZF_SIMT_FN void lds_wr128(bool p, uint32_t w, u32x4 v) const { if (p) *reinterpret_cast<u32x4*>(&lds[w]) = v; }
    // three 8-byte reads at ANY byte offset of the LDS array (gfx950 executes misaligned DS accesses; hipcc would split them into
    // bytes, hence the asm — which also waits for its own data, the compiler does not count asm memory operations)
    LZF_SIMT_FN void lds_rd64b3(bool p, uint32_t b0, uint32_t b1, uint32_t b2, uint64_t& v0, uint64_t& v1, uint64_t& v2) const {
        if (p) {
            uint64_t r0, r1, r2;
            asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %4\n\tds_read_b64 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(lds_a + b0), "v"(lds_a + b1), "v"(lds_a + b2) : "memory");
            v0 = r0; v1 = r1; v2 = r2;
        }
    }
    // ---- round 5: what a TEAM of wavefronts per block needs (lz4_compress_team.inc): the wave's index in the workgroup, values of a
    //      uniform lane, byte-addressed LDS reads at any alignment with several in flight, the min-lane tag protocol as one step,
    //      16-byte LDS accesses, flag words between the waves of a workgroup (explicit ds_* instructions: a volatile LDS pointer is a
    //      generic pointer to hipcc, i.e. FLAT accesses), a workgroup barrier
    LZF_SIMT_FN uint32_t wave() const { return threadIdx.x >> 6; }
    LZF_SIMT_FN uint32_t readlane(uint32_t v, uint32_t src) const { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src); }   // src uniform
    LZF_SIMT_FN uint64_t lds_rd64bu(uint32_t byte) const {
        uint64_t r; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_a + byte) : "memory"); return r;
    }
    LZF_SIMT_FN uint32_t lds_rd32b(bool p, uint32_t byte) const {
        uint32_t r = 0; if (p) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_a + byte) : "memory"); return r;
    }
    LZF_SIMT_FN void lds_rd8x2u(uint32_t b0, uint32_t b1, uint32_t& v0, uint32_t& v1) const {
        asm volatile("ds_read_u8 %0, %2\n\tds_read_u8 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(lds_a + b0), "v"(lds_a + b1) : "memory");
    }
    LZF_SIMT_FN uint32_t lds_rd8(bool p, uint32_t byte) const {
        uint32_t r = 0; if (p) asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_a + byte) : "memory"); return r;
    }
    // old = lds[w]; lds[w] = MARK; lds[w] = min(lds[w], tag); first = lds[w]  (one wave's LDS accesses execute in order)
    LZF_SIMT_FN void lds_tag(uint32_t w, uint32_t tag, uint32_t& old, uint32_t& first) const {
        asm volatile("ds_read_b32 %0, %2\n\tds_write_b32 %2, %3\n\tds_min_u32 %2, %4\n\tds_read_b32 %1, %2\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(old), "=&v"(first) : "v"(lds_a + 4u * w), "v"(0xFFFFFFFFu), "v"(tag) : "memory");
    }
    LZF_SIMT_FN void lds_wr32a(uint32_t w, uint32_t v) const { asm volatile("ds_write_b32 %0, %1" ::"v"(lds_a + 4u * w), "v"(v) : "memory"); }
    LZF_SIMT_FN u32x4 lds_rd128(uint32_t w) const {
        u32x4 r; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_a + 4u * w) : "memory"); return r;
    }
    LZF_SIMT_FN void lds_wr128a(bool p, uint32_t w, u32x4 v) const { if (p) asm volatile("ds_write_b128 %0, %1" ::"v"(lds_a + 4u * w), "v"(v) : "memory"); }
    LZF_SIMT_FN uint32_t flag_rd(uint32_t w) const {
        uint32_t r; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_a + 4u * w) : "memory"); return rfl(r);
    }
    LZF_SIMT_FN void flag_rd2(uint32_t w, uint32_t& a, uint32_t& c) const {          // two adjacent flag words, w even
        uint64_t r; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_a + 4u * w) : "memory");
        a = rfl((uint32_t)r); c = rfl((uint32_t)(r >> 32));
    }
    LZF_SIMT_FN void flag_wr(uint32_t w, uint32_t v) const { if (lane() == 0u) asm volatile("ds_write_b32 %0, %1" ::"v"(lds_a + 4u * w), "v"(v) : "memory"); }
    LZF_SIMT_FN void flag_wr2(uint32_t w, uint32_t v0, uint32_t v1) const {
        if (lane() == 0u) asm volatile("ds_write_b64 %0, %1" ::"v"(lds_a + 4u * w), "v"(((uint64_t)v1 << 32) | v0) : "memory");
    }
    // ---- the searcher's straight path (lz4_compress_team.inc "turbo"): byte offsets into the LDS array
    // two bytes and 8 bytes at any alignment, one round trip
    LZF_SIMT_FN void lds_rd8x2_64u(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t& v0, uint32_t& v1, uint64_t& v2) const {
        asm volatile("ds_read_u8 %0, %3\n\tds_read_u8 %1, %4\n\tds_read_b64 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(lds_a + b0), "v"(lds_a + b1), "v"(lds_a + b2) : "memory");
    }
    // LDS byte ADDRESS of the U32Table slot of the 8 bytes v8 (mod.rs:41-51; the arithmetic of simt_hash5), the table at the 16 KiB-aligned
    // address tab_addr (a VGPR: VOP3 takes one scalar operand on gfx9): the slot's byte offset is bits 26..39 of v * K with the low two cleared,
    // OR-ed into tab_addr — which is why the kernel's LDS array is declared aligned(16384) and checks its base at entry
    LZF_SIMT_FN uint32_t hash5_slot_addr(uint64_t v8, uint32_t tab_addr) const {
        const uint32_t xl = (uint32_t)v8, xh = (uint32_t)(v8 >> 32);
        uint32_t t, u, r; uint64_t p;
        asm("v_and_b32 %0, 0xff, %3\n\tv_mul_u32_u24 %0, 0xcf, %0\n\tv_and_b32 %1, 0xff, %4\n\tv_mad_u32_u24 %0, %1, %5, %0\n\t"
            "v_mad_u64_u32 %2, vcc, %3, %6, 0"
            : "=&v"(t), "=&v"(u), "=&v"(p) : "v"(xl), "v"(xh), "s"(0xBBu), "s"(0x1BBCDCBBu) : "vcc");
        const uint32_t hi = (uint32_t)(p >> 32) + t;
        asm("v_alignbit_b32 %0, %1, %2, 26\n\tv_and_or_b32 %0, %0, %3, %4" : "=&v"(r) : "v"(hi), "v"((uint32_t)p), "s"(0x3FFCu), "v"(tab_addr));
        return r;
    }
    LZF_SIMT_FN uint32_t lds_byte_addr(uint32_t byte) const { return lds_a + byte; }
    // the tag protocol and plain stores on byte ADDRESSES (hash5_slot_addr / lds_byte_addr)
    LZF_SIMT_FN void lds_tag_at(uint32_t addr, uint32_t tag, uint32_t& old, uint32_t& first) const {
        asm volatile("ds_read_b32 %0, %2\n\tds_write_b32 %2, %3\n\tds_min_u32 %2, %4\n\tds_read_b32 %1, %2\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(old), "=&v"(first) : "v"(addr), "v"(0xFFFFFFFFu), "v"(tag) : "memory");
    }
    LZF_SIMT_FN void lds_wr32_at(uint32_t addr, uint32_t v) const { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
    // lane 0 alone writes a 16-byte descriptor and, behind it, two adjacent flag words.  All 64 lanes must be active at the call
    // (the searcher's control flow is wave-uniform): EXEC is set to lane 0 and back to all lanes around the two stores.
    LZF_SIMT_FN void push16_flag2(uint32_t desc_byte, u32x4 d, uint32_t flag_byte, uint32_t f0, uint32_t f1) const {
        unsigned long long sv;      // (the incoming EXEC saved and restored, not assumed to be all lanes: ADVICE r5)
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b128 %1, %2\n\tds_write_b64 %3, %4\n\ts_mov_b64 exec, %0"
                     : "=&s"(sv) : "v"(lds_a + desc_byte), "v"(d), "v"(lds_a + flag_byte), "v"(((uint64_t)f1 << 32) | f0) : "memory");
    }
    // compare masks as scalars (v_cmp writes the lane mask directly; a ballot of a bool goes through a vector register and back), the first
    // set lane of a mask bounded from above in two scalar instructions (s_ff1 gives 0xFFFFFFFF for an empty mask), and a predicated
    // 4-byte read at any alignment that leaves the other lanes' value alone
    LZF_SIMT_FN unsigned long long mask_ne(uint32_t a, uint32_t c) const { return __builtin_amdgcn_uicmp(a, c, 33); }
    LZF_SIMT_FN unsigned long long mask_eq(uint32_t a, uint32_t c) const { return __builtin_amdgcn_uicmp(a, c, 32); }
    LZF_SIMT_FN unsigned long long mask_lt(uint32_t a, uint32_t c) const { return __builtin_amdgcn_uicmp(a, c, 36); }
    LZF_SIMT_FN uint32_t first_lane_min(unsigned long long m, uint32_t bound) const {
        uint32_t r; asm("s_ff1_i32_b64 %0, %1\n\ts_min_u32 %0, %0, %2" : "=&s"(r) : "s"(m), "s"(bound) : "scc"); return r;      // (early clobber: r must not share bound's register)
    }
    // per-lane select by a UNIFORM lane mask (an SGPR pair built by scalar instructions): a scalar result feeding a vector instruction costs
    // nothing, a vector compare feeding a scalar one ~16 cycles on a lone wave (tools/lone_wave_microbench.hip) — bounds that are uniform are
    // therefore turned into masks on the scalar side instead of being compared per lane
    LZF_SIMT_FN uint32_t sel(unsigned long long m, uint32_t if1, uint32_t if0) const {
        uint32_t r; asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(m)); return r;
    }
    LZF_SIMT_FN uint32_t lds_rd32bu(uint32_t byte) const {
        uint32_t r; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_a + byte) : "memory"); return r;
    }
    LZF_SIMT_FN uint32_t smin(uint32_t a, uint32_t c) const { uint32_t r; asm("s_min_u32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(c) : "scc"); return r; }      // (uniform operands: stays on the scalar side)
    LZF_SIMT_FN uint32_t ff1_32(uint32_t m) const { uint32_t r; asm("s_ff1_i32_b32 %0, %1" : "=s"(r) : "s"(m)); return r; }      // 0xFFFFFFFF for 0
    LZF_SIMT_FN uint32_t lds_rd32b_keep(bool p, uint32_t byte, uint32_t keep) const {
        uint32_t r = keep; if (p) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(lds_a + byte) : "memory"); return r;
    }
    LZF_SIMT_FN void sleep() const { __builtin_amdgcn_s_sleep(2); }
    LZF_SIMT_FN void barrier() const { __syncthreads(); }
    LZF_SIMT_FN void lds_fence() const { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    LZF_SIMT_FN uint32_t atomic_inc(uint32_t* q) const { return atomicAdd(q, 1u); }
    LZF_SIMT_FN uint64_t clock() const { return (uint64_t)clock64(); }
    LZF_SIMT_FN void keep(uint32_t v) const { asm volatile("" ::"v"(v)); }      // the value counts as used (a load made only to warm the caches)
    LZF_SIMT_FN uint64_t clock_fenced() const { __builtin_amdgcn_s_waitcnt(0); return (uint64_t)clock64(); }      // (debug timers) every outstanding access first
    static LZF_SIMT_FN void lds_mskor32_raw(uint32_t a, uint32_t mask, uint32_t val) { asm volatile("ds_mskor_b32 %0, %1, %2" ::"v"(a), "v"(mask), "v"(val) : "memory"); }
};
// mod.rs:41-51: v = 8 bytes LE, ((v << 24) * 889523592379) >> 52 — bits 28..39 of the low 40 bits of v * K.  With v = xl + (xh << 32),
// K = 0x1BBCDCBB + (0xCF << 32): v * K mod 2^40 = xl * 0x1BBCDCBB + ((xl * 0xCF + xh * 0xBB) mod 2^8 << 32): one 32 x 32 -> 64
// multiply-add and two 24-bit multiplies (full rate) instead of the three quarter-rate multiplies of a 64-bit product.  hipcc
// folds the C form of this back into the 64-bit multiply, hence the asm.
LZF_SIMT_FN uint32_t simt_hash5(uint64_t v8) {
    const uint32_t xl = (uint32_t)v8, xh = (uint32_t)(v8 >> 32);
    uint32_t t, u; uint64_t p;
    asm("v_and_b32 %0, 0xff, %1\n\tv_mul_u32_u24 %0, 0xcf, %0" : "=v"(t) : "v"(xl));
    asm("v_and_b32 %0, 0xff, %1\n\tv_mul_u32_u24 %0, 0xbb, %0" : "=v"(u) : "v"(xh));
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(xl), "s"(0x1BBCDCBBu) : "vcc");
    const uint32_t hi = (uint32_t)(p >> 32) + t + u;
    return ((hi & 0xFFu) << 4) | ((uint32_t)p >> 28);
}
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
static inline uint32_t simt_hash5(uint64_t v8) { return (uint32_t)(((v8 << 24) * 889523592379ull) >> 52); }      // mod.rs:41-51
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
    // round 5 (teams of wavefronts, tests/emu/emu_compress_team.cpp): the LDS the primitives address (this wave's own array or the
    // workgroup's), the wave's index, the next wave of the workgroup (a ring; a lone wave points at itself), barrier bookkeeping
    uint32_t* L;
    uint32_t wave_id;
    EmuWave* next;
    uint32_t* bar_count;            // arrivals at workgroup barriers so far (shared by the waves of a workgroup)
    uint32_t n_waves;
    uint32_t bar_gen;               // barriers this wave has passed
};
// the wave whose lanes are running (a fiber that starts reads its lane number from emu_current()->cur)
static inline EmuWave*& emu_current() { static EmuWave* w = nullptr; return w; }
struct SimtEmu {
    EmuWave* w;
    uint32_t my;
    uint32_t lane() const { return my; }
    // every lane reaches the same primitive before any lane goes on; behind the last lane of a wave the next wave of the workgroup
    // runs ITS lanes up to their next primitive (one fixed interleaving of the waves, deterministic)
    void sync() const {
        w->n_sync++;
        if (my + 1u < 64u) { w->cur = my + 1u; lzf_emu_switch(&w->sp[my], w->sp[my + 1u]); }
        else { EmuWave* nx = w->next; nx->cur = 0; emu_current() = nx; lzf_emu_switch(&w->sp[my], nx->sp[0]); }
    }
    unsigned long long ballot(bool p) const {
        sync(); w->xchg[my] = p ? 1u : 0u; sync();
        unsigned long long m = 0; for (uint32_t i = 0; i < 64; ++i) m |= (unsigned long long)(w->xchg[i] & 1u) << i;
        return m;
    }
    bool any(bool p) const { return ballot(p) != 0ull; }
    uint32_t bperm(uint32_t src_lane, uint32_t v) const { sync(); w->xchg[my] = v; sync(); return (uint32_t)w->xchg[src_lane & 63u]; }
    uint32_t lds_rd32(bool p, uint32_t i) const { sync(); return p ? w->L[i] : 0u; }
    void lds_wr32(bool p, uint32_t i, uint32_t v) const { sync(); if (p) w->L[i] = v; }
    void lds_min32(bool p, uint32_t i, uint32_t v) const { sync(); if (p && v < w->L[i]) w->L[i] = v; }
    void lds_mskor32(bool p, uint32_t i, uint32_t mask, uint32_t val) const { sync(); if (p) w->L[i] = (w->L[i] & ~mask) | val; }
    uint32_t lds_rd32u(uint32_t i) const { return lds_rd32(true, i); }
    void lds_wr32u(uint32_t i, uint32_t v) const { lds_wr32(true, i, v); }
    void lds_min32u(uint32_t i, uint32_t v) const { lds_min32(true, i, v); }
    void lds_mskor32u(uint32_t i, uint32_t mask, uint32_t val) const { lds_mskor32(true, i, mask, val); }
    void lds_wr128(bool p, uint32_t i, u32x4 v) const { sync(); if (p) memcpy(&w->L[i], &v, 16); }
    void lds_rd64b3(bool p, uint32_t b0, uint32_t b1, uint32_t b2, uint64_t& v0, uint64_t& v1, uint64_t& v2) const {
        sync();
        if (p) { const uint8_t* l = (const uint8_t*)w->L; memcpy(&v0, l + b0, 8); memcpy(&v1, l + b1, 8); memcpy(&v2, l + b2, 8); }
    }
    // ---- round 5 (see SimtGpu)
    uint32_t wave() const { return w->wave_id; }
    uint32_t readlane(uint32_t v, uint32_t src) const { return bperm(src, v); }
    uint64_t lds_rd64bu(uint32_t byte) const { sync(); uint64_t v; memcpy(&v, (const uint8_t*)w->L + byte, 8); return v; }
    uint32_t lds_rd32b(bool p, uint32_t byte) const { sync(); uint32_t v = 0; if (p) memcpy(&v, (const uint8_t*)w->L + byte, 4); return v; }
    void lds_rd8x2u(uint32_t b0, uint32_t b1, uint32_t& v0, uint32_t& v1) const { sync(); const uint8_t* l = (const uint8_t*)w->L; v0 = l[b0]; v1 = l[b1]; }
    uint32_t lds_rd8(bool p, uint32_t byte) const { sync(); return p ? ((const uint8_t*)w->L)[byte] : 0u; }
    void lds_tag(uint32_t i, uint32_t tag, uint32_t& old, uint32_t& first) const {
        old = lds_rd32(true, i); lds_wr32(true, i, 0xFFFFFFFFu); lds_min32(true, i, tag); first = lds_rd32(true, i);
    }
    void lds_wr32a(uint32_t i, uint32_t v) const { lds_wr32(true, i, v); }
    u32x4 lds_rd128(uint32_t i) const { sync(); u32x4 v; memcpy(&v, &w->L[i], 16); return v; }
    void lds_wr128a(bool p, uint32_t i, u32x4 v) const { lds_wr128(p, i, v); }
    uint32_t flag_rd(uint32_t i) const { sync(); return w->L[i]; }
    void flag_rd2(uint32_t i, uint32_t& a, uint32_t& c) const { sync(); a = w->L[i]; c = w->L[i + 1u]; }
    void flag_wr(uint32_t i, uint32_t v) const { sync(); if (my == 0u) w->L[i] = v; }
    void flag_wr2(uint32_t i, uint32_t v0, uint32_t v1) const { sync(); if (my == 0u) { w->L[i] = v0; w->L[i + 1u] = v1; } }
    void lds_rd8x2_64u(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t& v0, uint32_t& v1, uint64_t& v2) const {
        sync(); const uint8_t* l = (const uint8_t*)w->L; v0 = l[b0]; v1 = l[b1]; memcpy(&v2, l + b2, 8);
    }
    uint32_t hash5_slot_addr(uint64_t v8, uint32_t tab_addr) const { return tab_addr + 4u * simt_hash5(v8); }
    uint32_t lds_byte_addr(uint32_t byte) const { return byte; }
    void lds_tag_at(uint32_t addr, uint32_t tag, uint32_t& old, uint32_t& first) const { lds_tag(addr >> 2, tag, old, first); }
    void lds_wr32_at(uint32_t addr, uint32_t v) const { lds_wr32(true, addr >> 2, v); }
    void push16_flag2(uint32_t desc_byte, u32x4 d, uint32_t flag_byte, uint32_t f0, uint32_t f1) const {
        sync(); if (my == 0u) { memcpy((uint8_t*)w->L + desc_byte, &d, 16); w->L[flag_byte >> 2] = f0; w->L[(flag_byte >> 2) + 1u] = f1; }
    }
    unsigned long long mask_ne(uint32_t a, uint32_t c) const { return ballot(a != c); }
    unsigned long long mask_eq(uint32_t a, uint32_t c) const { return ballot(a == c); }
    unsigned long long mask_lt(uint32_t a, uint32_t c) const { return ballot(a < c); }
    uint32_t first_lane_min(unsigned long long m, uint32_t bound) const { const uint32_t f = m ? simt_ctz64(m) : 0xFFFFFFFFu; return f < bound ? f : bound; }
    uint32_t sel(unsigned long long m, uint32_t if1, uint32_t if0) const { return ((m >> my) & 1ull) ? if1 : if0; }
    uint32_t lds_rd32bu(uint32_t byte) const { sync(); uint32_t v; memcpy(&v, (const uint8_t*)w->L + byte, 4); return v; }
    uint32_t smin(uint32_t a, uint32_t c) const { return a < c ? a : c; }
    uint32_t ff1_32(uint32_t m) const { return m ? simt_ctz32(m) : 0xFFFFFFFFu; }
    uint32_t lds_rd32b_keep(bool p, uint32_t byte, uint32_t keep) const { sync(); uint32_t v = keep; if (p) memcpy(&v, (const uint8_t*)w->L + byte, 4); return v; }
    void sleep() const { sync(); }
    void lds_fence() const {}
    // all waves of the workgroup: the lanes of a wave test the counter at the same lock-step point (other waves only run between
    // two rotations of this wave), so they leave together
    void barrier() const {
        sync();
        if (my == 0u) { ++*w->bar_count; ++w->bar_gen; }
        sync();
        for (;;) { const bool ok = *w->bar_count >= w->bar_gen * w->n_waves; sync(); if (ok) break; }
    }
    uint32_t atomic_inc(uint32_t* q) const { return (*q)++; }
    uint64_t clock() const { return 0; }
    void keep(uint32_t) const {}
    uint64_t clock_fenced() const { return 0; }
};
}  // namespace lzf
#endif
