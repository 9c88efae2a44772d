// tools/lds_width_microbench.hip — cost and correctness of misaligned DS accesses by width on gfx950 (a lone wavefront, 13 or 64 lanes):
// one read + wait + one write per round.  ANALYSIS TOOL.   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_w tools/lds_width_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int W, int MIS>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint32_t rounds, unsigned long long ml) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[65536];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 65536 / 4; i += 64) ((uint32_t*)ring)[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)ring;
    const uint32_t sa = base + (((lane * 1237u + 13u) & 0x3FFFu) & ~15u) + MIS, da = base + 0x8000u + (((lane * 977u + 5u) & 0x3FFFu) & ~15u) + (MIS ? MIS + 2 : 0);
    // correctness first: copy W bytes from sa to da, compare bytewise
    uint32_t bad = 0;
    if (W == 4) asm volatile("ds_read_b32 v100, %0\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b32 %1, v100\n\ts_waitcnt lgkmcnt(0)" :: "v"(sa), "v"(da) : "memory", "v100");
    if (W == 8) asm volatile("ds_read_b64 v[100:101], %0\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b64 %1, v[100:101]\n\ts_waitcnt lgkmcnt(0)" :: "v"(sa), "v"(da) : "memory", "v100", "v101");
    if (W == 16) asm volatile("ds_read_b128 v[100:103], %0\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b128 %1, v[100:103]\n\ts_waitcnt lgkmcnt(0)" :: "v"(sa), "v"(da) : "memory", "v100", "v101", "v102", "v103");
    __syncthreads();
    for (int i = 0; i < W; ++i) if (ring[sa - base + i] != ring[da - base + i]) ++bad;
    const long long t0 = clock64();
    asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, %0" :: "s"(ml) : "s20", "s21");
    for (uint32_t r = 0; r < rounds; ++r) {
        if (W == 4) asm volatile("ds_read_b32 v100, %0\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b32 %1, v100" :: "v"(sa), "v"(da) : "memory", "v100");
        if (W == 8) asm volatile("ds_read_b64 v[100:101], %0\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b64 %1, v[100:101]" :: "v"(sa), "v"(da) : "memory", "v100", "v101");
        if (W == 16) asm volatile("ds_read_b128 v[100:103], %0\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b128 %1, v[100:103]" :: "v"(sa), "v"(da) : "memory", "v100", "v101", "v102", "v103");
    }
    asm volatile("s_mov_b64 exec, s[20:21]" ::: "s20", "s21");
    const long long t1 = clock64();
    const unsigned long long anybad = __ballot(bad != 0);
    if (lane == 0) { out[0] = (uint32_t)((t1 - t0) / rounds); out[1] = (uint32_t)__builtin_popcountll(anybad); }
}
int main() {
    uint32_t* d; (void)hipMalloc(&d, 64); uint32_t h[2]; const uint32_t R = 20000;
    for (unsigned long long ml : {~0ull, 0x8421084210842108ull}) {
        printf("lanes %2d:", __builtin_popcountll(ml));
#define RUN(W, MIS) hipLaunchKernelGGL((k<W, MIS>), dim3(1), dim3(64), 0, 0, d, R, ml); (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost); printf("  b%d%s %u cyc%s", W * 8, MIS ? "+mis" : "    ", h[0], h[1] ? " WRONG" : "");
        RUN(4, 0) RUN(4, 3) RUN(8, 0) RUN(8, 3) RUN(16, 0) RUN(16, 3)
        printf("\n");
    }
    return 0;
}
