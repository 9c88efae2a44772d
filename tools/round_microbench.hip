// tools/round_microbench.hip — what a resolver round costs on a lone wavefront (gfx950): the round as the segmented pipeline issues it
// (EXEC switched per class) against the same DS work under one EXEC mask, with and without the wait.  ANALYSIS TOOL.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/round_mb tools/round_microbench.hip && /tmp/round_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int V>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint32_t rounds, unsigned long long ml0, unsigned long long mA, unsigned long long m2, unsigned long long m3, unsigned long long m4) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[131072];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 131072 / 4; i += 64) ((uint32_t*)ring)[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)ring;
    uint32_t sa = base + ((lane * 1237u + 13u) & 0xFFFFu), da = base + 0x10000u + ((lane * 977u + 5u) & 0xFFFFu);
    const uint32_t M = 9u + (lane & 7u);
    unsigned long long ml = ml0;
    const long long t0 = clock64();
    for (uint32_t r = 0; r < rounds; ++r) {
        if (V == 0) {            // as issued today: EXEC per class
            asm volatile(
                "s_mov_b64 s[20:21], exec\n\t"
                "s_and_b64 exec, %[ml], %[mA]\n\t" "ds_read_b32 v100, %[sa]\n\t" "ds_read_b32 v101, %[s1]\n\t"
                "s_and_b64 exec, %[ml], %[m2]\n\t" "ds_read_b64 v[102:103], %[sa]\n\t" "ds_read_b64 v[104:105], %[s3]\n\t"
                "s_and_b64 exec, %[ml], %[m3]\n\t" "s_cbranch_execz Lr%=\n\t" "ds_read_b64 v[106:107], %[sa] offset:8\n\t" "ds_read_b64 v[108:109], %[s3] offset:8\n\t"
                "s_and_b64 exec, %[ml], %[m4]\n\t" "s_cbranch_execz Lr%=\n\t" "ds_read_b64 v[110:111], %[sa] offset:16\n\t" "ds_read_b64 v[112:113], %[sa] offset:24\n\t"
                "Lr%=:\n\t" "s_waitcnt lgkmcnt(0)\n\t"
                "s_and_b64 exec, %[ml], %[mA]\n\t" "ds_write_b32 %[da], v100\n\t" "ds_write_b32 %[d1], v101\n\t"
                "s_and_b64 exec, %[ml], %[m2]\n\t" "ds_write_b64 %[da], v[102:103]\n\t" "ds_write_b64 %[d3], v[104:105]\n\t"
                "s_and_b64 exec, %[ml], %[m3]\n\t" "s_cbranch_execz Lw%=\n\t" "ds_write_b64 %[da], v[106:107] offset:8\n\t" "ds_write_b64 %[d3], v[108:109] offset:8\n\t"
                "s_and_b64 exec, %[ml], %[m4]\n\t" "s_cbranch_execz Lw%=\n\t" "ds_write_b64 %[da], v[110:111] offset:16\n\t" "ds_write_b64 %[da], v[112:113] offset:24\n\t"
                "Lw%=:\n\t" "s_mov_b64 exec, s[20:21]\n\t"
                :: [ml] "s"(ml), [mA] "s"(mA), [m2] "s"(m2), [m3] "s"(m3), [m4] "s"(m4), [sa] "v"(sa), [s1] "v"(sa + M - 4), [s3] "v"(sa + M - 8), [da] "v"(da), [d1] "v"(da + M - 4), [d3] "v"(da + M - 8)
                : "memory", "s20", "s21", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113");
        } else if (V == 1 || V == 2) {     // one EXEC mask; the classes' presence tested on scalars (V == 2: without the wait)
            asm volatile(
                "s_mov_b64 s[20:21], exec\n\t" "s_mov_b64 exec, %[ml]\n\t"
                "ds_read_b32 v100, %[sa]\n\t" "ds_read_b32 v101, %[s1]\n\t" "ds_read_b64 v[102:103], %[sa]\n\t" "ds_read_b64 v[104:105], %[s3]\n\t"
                "s_and_b64 s[22:23], %[ml], %[m3]\n\t" "s_cbranch_scc0 Lr%=\n\t" "ds_read_b64 v[106:107], %[sa] offset:8\n\t" "ds_read_b64 v[108:109], %[s3] offset:8\n\t"
                "s_and_b64 s[24:25], %[ml], %[m4]\n\t" "s_cbranch_scc0 Lr%=\n\t" "ds_read_b64 v[110:111], %[sa] offset:16\n\t" "ds_read_b64 v[112:113], %[sa] offset:24\n\t"
                "Lr%=:\n\t"
                :: [ml] "s"(ml), [m3] "s"(m3), [m4] "s"(m4), [sa] "v"(sa), [s1] "v"(sa + M - 4), [s3] "v"(sa + M - 8)
                : "memory", "s20", "s21", "s22", "s23", "s24", "s25", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113");
            if (V == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile(
                "ds_write_b32 %[da], v100\n\t" "ds_write_b32 %[d1], v101\n\t" "ds_write_b64 %[da], v[102:103]\n\t" "ds_write_b64 %[d3], v[104:105]\n\t"
                "s_cmp_lg_u64 s[22:23], 0\n\t" "s_cbranch_scc0 Lw%=\n\t" "ds_write_b64 %[da], v[106:107] offset:8\n\t" "ds_write_b64 %[d3], v[108:109] offset:8\n\t"
                "s_cmp_lg_u64 s[24:25], 0\n\t" "s_cbranch_scc0 Lw%=\n\t" "ds_write_b64 %[da], v[110:111] offset:16\n\t" "ds_write_b64 %[da], v[112:113] offset:24\n\t"
                "Lw%=:\n\t" "s_mov_b64 exec, s[20:21]\n\t"
                :: [da] "v"(da), [d1] "v"(da + M - 4), [d3] "v"(da + M - 8)
                : "memory", "s20", "s21", "s22", "s23", "s24", "s25", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113");
        } else if (V == 3) {     // only the A + 2 classes, one EXEC mask: the floor of a round (4 reads, wait, 4 writes)
            asm volatile(
                "ds_read_b32 v100, %[sa]\n\t" "ds_read_b32 v101, %[s1]\n\t" "ds_read_b64 v[102:103], %[sa]\n\t" "ds_read_b64 v[104:105], %[s3]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "ds_write_b32 %[da], v100\n\t" "ds_write_b32 %[d1], v101\n\t" "ds_write_b64 %[da], v[102:103]\n\t" "ds_write_b64 %[d3], v[104:105]\n\t"
                :: [sa] "v"(sa), [s1] "v"(sa + M - 4), [s3] "v"(sa + M - 8), [da] "v"(da), [d1] "v"(da + M - 4), [d3] "v"(da + M - 8)
                : "memory", "v100", "v101", "v102", "v103", "v104", "v105");
        } else if (V == 4) {     // one read, wait, one write: the LDS round trip itself
            asm volatile("ds_read_b64 v[102:103], %[sa]\n\t" "s_waitcnt lgkmcnt(0)\n\t" "ds_write_b64 %[da], v[102:103]\n\t"
                :: [sa] "v"(sa), [da] "v"(da) : "memory", "v102", "v103");
        }
        // a level loop's scalar part, so that rounds do not fuse: rotate the masks
        ml = (ml << 1) | (ml >> 63);
    }
    const long long t1 = clock64();
    if (lane == 0) out[0] = (uint32_t)((t1 - t0) / rounds);
}
int main() {
    uint32_t* d; hipMalloc(&d, 64); uint32_t h = 0; const uint32_t R = 20000;
    struct { const char* name; unsigned long long ml_density; unsigned long long mA, m2, m3, m4; } cfg[] = {
        {"all four classes present", 0, 0x1111111111111111ull, 0xEEEEEEEEEEEEEEEEull, 0xCCCCCCCCCCCCCCCCull, 0x8888888888888888ull},
        {"classes A, 2, 3         ", 0, 0x1111111111111111ull, 0xEEEEEEEEEEEEEEEEull, 0xCCCCCCCCCCCCCCCCull, 0},
        {"classes A, 2            ", 0, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull, 0, 0}};
    for (unsigned long long ml0 : {~0ull, 0x8421084210842108ull /* 12 lanes */, 0x0001000100010001ull /* 4 lanes */}) for (auto& c : cfg) {
        printf("lanes %2d | ", __builtin_popcountll(ml0));
        printf("%s:", c.name);
#define RUN(V) hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d, R, ml0, c.mA, c.m2, c.m3, c.m4); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("  V%d %u", V, h);
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
        printf("  cycles/round\n");
    }
    return 0;
}
