// lone_wave_microbench.hip — what does ONE wavefront alone on its SIMD pay per dependent instruction and per LDS round trip on gfx950?
// (the searcher of lzf_compress_team_kernel is exactly that: profiles/r05_team_kernel_block_counters.txt)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lone_wave_microbench tools/lone_wave_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
__global__ void k(uint64_t* out, uint32_t seed, uint32_t which) {
    __shared__ uint32_t lds[16384];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 16384; i += 64) lds[i] = (i * 2654435761u) >> 18;      // values < 16384
    __syncthreads();
    uint32_t v = seed + lane, w = lane * 4u, a = 0; uint64_t t0, t1;
    // 1: dependent VALU chain
    if (which & 1u) { t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) { asm volatile(REP64("v_add_u32 %0, %0, %1\n\t") : "+v"(v) : "v"(w)); }
    t1 = __builtin_readcyclecounter(); if (lane == 0) out[0] = t1 - t0; }
    // 2: four independent VALU chains
    if (which >> 1 & 1u) { uint32_t x0 = v, x1 = v + 1, x2 = v + 2, x3 = v + 3;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(w)); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[1] = t1 - t0; v ^= x0 ^ x1 ^ x2 ^ x3; }
    // 3: dependent SALU chain
    if (which >> 2 & 1u) { uint32_t s = seed;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP64("s_add_u32 %0, %0, 3\n\t") : "+s"(s) :: "scc"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[2] = t1 - t0; v ^= s; }
    // 4: v_cmp -> s_ff1 -> v_readlane -> v_add chain (VALU <-> SALU crossings)
    if (which >> 3 & 1u) { uint32_t s = 0;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("v_cmp_ne_u32 vcc, %0, %2\n\ts_ff1_i32_b64 %1, vcc\n\ts_and_b32 %1, %1, 63\n\ts_nop 3\n\tv_readlane_b32 %1, %0, %1\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v), "+s"(s) : "v"(w) : "vcc", "scc"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[3] = t1 - t0; v ^= s; }
    // 5: dependent aligned ds_read_b32 chain (address = value read), 64 lanes
    if (which >> 4 & 1u) { uint32_t ad = (lane * 64u) & 0xFFFCu;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_lshlrev_b32 %0, 2, %0\n\t") : "+v"(ad) :: "memory"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[4] = t1 - t0; v ^= ad; }
    // 6: the same with 16 active lanes
    if (which >> 5 & 1u) { uint32_t ad = (lane * 64u) & 0xFFFCu;
      t0 = __builtin_readcyclecounter();
      if (lane < 16) for (int it = 0; it < 64; ++it) { asm volatile(REP16("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_lshlrev_b32 %0, 2, %0\n\t") : "+v"(ad) :: "memory"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[5] = t1 - t0; v ^= ad; }
    // 7: misaligned ds_read_b64, 64 lanes, dependent
    if (which >> 6 & 1u) { uint32_t ad = lane * 67u + 1u; uint64_t r = 0;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) {
#pragma unroll
          for (int u = 0; u < 16; ++u) { asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(ad) : "memory"); ad = (uint32_t)r & 0x7fffu; }
      }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[6] = t1 - t0; v ^= ad; }
    // 8: the min-lane tag protocol (read, write, min, read) on 16 active slots, dependent through the address
    if (which >> 7 & 1u) { uint32_t ad = (lane * 64u) & 0xFFFCu, o = 0, f = 0;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("ds_read_b32 %1, %0\n\tds_write_b32 %0, %3\n\tds_min_u32 %0, %4\n\tds_read_b32 %2, %0\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1\n\tv_lshlrev_b32 %0, 2, %1\n\t") : "+v"(ad), "+v"(o), "+v"(f) : "v"(0xFFFFFFFFu), "v"(lane) : "memory"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[7] = t1 - t0; v ^= ad ^ f; }
    // 9: two ds_read_u8 + one misaligned ds_read_b64 in one wait (the measurement trip), dependent
    if (which >> 8 & 1u) { uint32_t ad = lane * 67u + 1u, x = 0, y = 0; uint64_t r = 0;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) {
#pragma unroll
          for (int u = 0; u < 16; ++u) {
              asm volatile("ds_read_u8 %1, %3\n\tds_read_u8 %2, %3 offset:7\n\tds_read_b64 %0, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r), "=&v"(x), "=&v"(y) : "v"(ad) : "memory");
              ad = (x ^ y ^ (uint32_t)r) & 0x7fffu;
          }
      }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[8] = t1 - t0; v ^= ad; }
    // 10..15: where the VALU <-> SALU crossings of (4) go
    if (which >> 9 & 1u) { uint32_t sreg = 0;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("v_cmp_ne_u32 vcc, %0, %2\n\ts_ff1_i32_b64 %1, vcc\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v), "+s"(sreg) : "v"(w) : "vcc"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[9] = t1 - t0; v ^= sreg; }
    if (which >> 10 & 1u) { uint32_t sreg = 0;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("v_readlane_b32 %1, %0, 5\n\ts_nop 1\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v), "+s"(sreg)); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[10] = t1 - t0; v ^= sreg; }
    if (which >> 11 & 1u) { uint32_t sreg = 0;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("v_readfirstlane_b32 %1, %0\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v), "+s"(sreg) :: "scc"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[11] = t1 - t0; v ^= sreg; }
    if (which >> 12 & 1u) { t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP64("s_nop 3\n\t")); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[12] = t1 - t0; }
    if (which >> 13 & 1u) { t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("v_cmp_ne_u32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_add_u32 %0, 1, %0\n\t") : "+v"(v) : "v"(w) : "vcc"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[13] = t1 - t0; }
    if (which >> 14 & 1u) { uint32_t sreg = 3;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("s_add_u32 %1, %1, 1\n\tv_add_u32 %0, %1, %0\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v), "+s"(sreg) :: "scc"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[14] = t1 - t0; v ^= sreg; }
    if (which >> 15 & 1u) { uint32_t sreg = 3;       // s_and_saveexec / restore around a VALU op
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("v_cmp_ne_u32 vcc, %0, %1\n\ts_and_saveexec_b64 s[20:21], vcc\n\tv_add_u32 %0, 1, %0\n\ts_mov_b64 exec, s[20:21]\n\t") : "+v"(v) : "v"(w) : "vcc", "s20", "s21", "scc"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[15] = t1 - t0; v ^= sreg; }
    if (which >> 16 & 1u) { uint32_t sl = 3, st = 0;      // SALU writes the lane select of a v_readlane
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("s_and_b32 %0, %0, 63\n\tv_readlane_b32 %1, %2, %0\n\ts_add_u32 %0, %0, %1\n\t") : "+s"(sl), "+s"(st) : "v"(v) : "scc"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[16] = t1 - t0; v ^= sl; }
    if (which >> 17 & 1u) { uint32_t sl = 3, st = 0;      // the same with the lane select a constant
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("s_and_b32 %0, %0, 63\n\tv_readlane_b32 %1, %2, 5\n\ts_add_u32 %0, %0, %1\n\t") : "+s"(sl), "+s"(st) : "v"(v) : "scc"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[17] = t1 - t0; v ^= sl; }
    if (which >> 18 & 1u) { uint32_t sl = 3;              // v_cmp -> s_ff1 -> s_min -> s_add chain back into a VALU compare operand
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("v_cmp_gt_u32 vcc, %0, %1\n\ts_ff1_i32_b64 %0, vcc\n\ts_min_u32 %0, %0, 40\n\t") : "+s"(sl) : "v"(lane) : "vcc", "scc"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[18] = t1 - t0; v ^= sl; }
    if (which >> 19 & 1u) { uint32_t sl = 3;              // SALU result consumed by a VALU, whose result goes back through v_readfirstlane
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("s_add_u32 %1, %1, 1\n\tv_add_u32 %0, %1, %0\n\tv_readfirstlane_b32 %1, %0\n\t") : "+v"(v), "+s"(sl) :: "scc"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[19] = t1 - t0; v ^= sl; }
    if (which >> 20 & 1u) {       // taken branches
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("s_branch 1f\n\ts_nop 0\n1:\n\ts_branch 2f\n\ts_nop 0\n2:\n\ts_branch 3f\n\ts_nop 0\n3:\n\ts_branch 4f\n\ts_nop 0\n4:\n\t")); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[20] = t1 - t0; }
    if (which >> 21 & 1u) {       // conditional branches not taken
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile("s_cmp_eq_u32 0, 1\n\t" REP64("s_cbranch_scc1 9f\n\t") "9:\n\t" ::: "scc"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[21] = t1 - t0; }
    if (which >> 22 & 1u) {       // dependent v_mad_u64_u32
      uint64_t acc = v;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) {
#pragma unroll
          for (int u = 0; u < 16; ++u) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(acc) : "v"((uint32_t)acc), "s"(0x1BBCDCBBu) : "vcc"); }
      }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[22] = t1 - t0; v ^= (uint32_t)acc; }
    if (which >> 23 & 1u) {       // ds_write_b32 issue, no wait in between (64 lanes, own words)
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("ds_write_b32 %0, %1\n\t") "s_waitcnt lgkmcnt(0)" :: "v"(lane * 4u), "v"(v) : "memory"); }
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[23] = t1 - t0; }
    if (which >> 24 & 1u) {       // the descriptor push: EXEC to lane 0, ds_write_b128 + ds_write_b64, EXEC back (no wait)
      typedef unsigned int u4 __attribute__((ext_vector_type(4)));
      u4 d = {v, v, v, v}; uint64_t fl = v;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < 64; ++it) { asm volatile(REP16("s_mov_b64 exec, 1\n\tds_write_b128 %0, %1\n\tds_write_b64 %0, %2 offset:64\n\ts_mov_b64 exec, -1\n\t") :: "v"(lane * 16u), "v"(d), "v"(fl) : "memory"); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      t1 = __builtin_readcyclecounter(); if (lane == 0) out[24] = t1 - t0; }
    if (lane == 0) out[31] = v + a;
}
int main(int argc, char** argv) {
    const unsigned mask = argc > 1 ? (unsigned)strtoul(argv[1], 0, 0) : 0xFFFFFu;
    uint64_t* d; hipMalloc(&d, 32 * 8); hipMemset(d, 0, 32 * 8); uint64_t h[32];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 12345u, mask); hipDeviceSynchronize(); }
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"dependent v_add_u32", "4 independent v_add chains", "dependent s_add_u32", "v_cmp/s_ff1/s_and/s_nop 3/v_readlane/v_add (6 instr)",
                           "dependent aligned ds_read_b32 (64 lanes) + shift", "the same, 16 lanes", "dependent misaligned ds_read_b64 (64 lanes) + and",
                           "tag protocol: 4 DS + wait + restore + shift (16 slots)", "2 x ds_read_u8 + misaligned ds_read_b64, one wait + 3 VALU",
                           "v_cmp -> vcc -> s_ff1 -> v_add (3 instr)", "v_readlane -> s_nop 1 -> v_add (3)", "v_readfirstlane -> s_add -> v_add (3)", "s_nop 3",
                           "v_cmp -> s_nop 1 -> v_cndmask vcc -> v_add (4)", "s_add -> v_add -> v_add (3)", "v_cmp -> s_and_saveexec -> v_add -> s_mov exec (4)",
                           "s_and -> v_readlane (lane select from the SALU) -> s_add (3)", "s_and -> v_readlane (constant lane) -> s_add (3)", "v_cmp -> s_ff1 -> s_min (3)", "s_add -> v_add -> v_readfirstlane (3)",
                           "taken s_branch (over one s_nop)", "s_cbranch_scc1 not taken", "dependent v_mad_u64_u32", "ds_write_b32 x16, one wait", "push: exec=1, ds_write_b128, ds_write_b64, exec=-1"};
    const double per[] = {4096, 4096, 4096, 1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024, 4096, 1024, 1024, 1024, 1024, 1024, 1024, 1024, 4096, 4096, 1024, 1024, 1024};
    printf("one wavefront alone on a CU, gfx950; __builtin_readcyclecounter() ticks per unit\n");
    for (int i = 0; i < 25; ++i) if (mask >> i & 1u) printf("%-60s %8.1f ticks per %s\n", names[i], (double)h[i] / per[i], (i < 3 || i == 12 || i == 20 || i == 21 || i == 22) ? "instruction" : "group");
    return 0;
}
