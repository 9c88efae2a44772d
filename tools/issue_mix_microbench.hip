// Issue-rate microbenchmark (analysis tool): how many wave-instructions per cycle a CU retires for
// dependent SALU / VALU chains and their mix, at a given number of resident waves per CU.
// hipcc --offload-arch=gfx950 -O3 tools/issue_mix_microbench.hip -o /tmp/issue_mix && /tmp/issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t* out, int iters, int lds_pad) {
    extern __shared__ uint32_t pad[];
    uint32_t v = threadIdx.x, s = blockIdx.x;
    if (lds_pad < 0) pad[threadIdx.x] = v;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // 16 dependent VALU
            asm volatile(
                "v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n"
                "v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n"
                "v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n"
                "v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n" : "+v"(v));
        } else if (MODE == 1) { // 16 dependent SALU
            asm volatile(
                "s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n"
                "s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n"
                "s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n"
                "s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n" : "+s"(s) : : "scc");
        } else if (MODE == 2) { // 8 SALU + 8 VALU interleaved, each chain dependent on itself
            asm volatile(
                "s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n v_xor_b32 %0, %0, 3\n"
                "s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n v_xor_b32 %0, %0, 3\n"
                "s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n v_xor_b32 %0, %0, 3\n"
                "s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n v_xor_b32 %0, %0, 3\n" : "+v"(v), "+s"(s) : : "scc");
        } else if (MODE == 3) { // exec-mask style: v_cmp -> s_and_saveexec -> valu -> s_or exec  (x4)
            asm volatile(
                "v_cmp_lt_u32 vcc, 7, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, 1\n s_or_b64 exec, exec, s[20:21]\n"
                "v_cmp_lt_u32 vcc, 7, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, 1\n s_or_b64 exec, exec, s[20:21]\n"
                "v_cmp_lt_u32 vcc, 7, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, 1\n s_or_b64 exec, exec, s[20:21]\n"
                "v_cmp_lt_u32 vcc, 7, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, 1\n s_or_b64 exec, exec, s[20:21]\n" : "+v"(v) : : "vcc", "s20", "s21");
        } else if (MODE == 4) { // 16 independent-ish VALU (4 chains)
            uint32_t a = v, b = v + 1, c = v + 2, d = v + 3;
            asm volatile(
                "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                "v_xor_b32 %0, %0, 3\n v_xor_b32 %1, %1, 3\n v_xor_b32 %2, %2, 3\n v_xor_b32 %3, %3, 3\n"
                "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                "v_xor_b32 %0, %0, 3\n v_xor_b32 %1, %1, 3\n v_xor_b32 %2, %2, 3\n v_xor_b32 %3, %3, 3\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            v = a ^ b ^ c ^ d;
        } else if (MODE == 5) { // readlane/ballot style: v_cmp to sgpr pair, s_ff1, v_readlane
            asm volatile(
                "v_cmp_lt_u32 s[20:21], 7, %0\n s_ff1_i32_b64 s22, s[20:21]\n v_readlane_b32 s23, %0, s22\n v_add_u32 %0, s23, %0\n"
                "v_cmp_lt_u32 s[20:21], 7, %0\n s_ff1_i32_b64 s22, s[20:21]\n v_readlane_b32 s23, %0, s22\n v_add_u32 %0, s23, %0\n"
                "v_cmp_lt_u32 s[20:21], 7, %0\n s_ff1_i32_b64 s22, s[20:21]\n v_readlane_b32 s23, %0, s22\n v_add_u32 %0, s23, %0\n"
                "v_cmp_lt_u32 s[20:21], 7, %0\n s_ff1_i32_b64 s22, s[20:21]\n v_readlane_b32 s23, %0, s22\n v_add_u32 %0, s23, %0\n" : "+v"(v) : : "s20", "s21", "s22", "s23");
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = v + s;
}

template <int MODE>
int run(const char* name, int waves_per_cu, uint32_t* d_out) {
    const int iters = 20000, grid = 256 * waves_per_cu;
    const int lds = (160 * 1024) / waves_per_cu > 65536 ? 65536 : ((160 * 1024) / waves_per_cu) & ~255;   // caps residency at waves_per_cu
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), lds, 0, d_out, 100, 0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), lds, 0, d_out, iters, 0);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    const double insts = 16.0 * iters * grid;
    printf("%-28s waves/CU %2d: %8.3f ms  %.2f Ginst/s  %.3f inst/ns/CU\n", name, waves_per_cu, ms, insts / ms / 1e6, insts / ms / 1e6 / 256);
    return 0;
}

int main() {
    uint32_t* d_out; CHECK(hipMalloc(&d_out, 256 * 32 * 64 * 4));
    for (int w : {4, 8, 12, 20, 32}) {
        run<0>("VALU dependent", w, d_out);
        run<4>("VALU 4 chains", w, d_out);
        run<1>("SALU dependent", w, d_out);
        run<2>("SALU+VALU interleaved", w, d_out);
        run<3>("cmp/saveexec/valu/or-exec", w, d_out);
        run<5>("cmp/ff1/readlane/valu", w, d_out);
    }
    return 0;
}
