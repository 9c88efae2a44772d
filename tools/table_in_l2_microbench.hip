// Feasibility probe for a compress kernel whose position table lives in HBM/L2 instead of LDS
// (LDS-free => 32 waves/CU instead of 10).  Each wave mimics one search step of lz4_compress.hip:
//   16 lanes gather 4-byte slots of a private 16 KiB table (sc1 loads: L2-served),
//   a dependent 16-lane gather of 16 bytes from a private 64 KiB window,
//   16 scattered 4-byte stores into the table.
// Reports steps/s for the whole chip at a given number of resident waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ __launch_bounds__(64) void probe(uint32_t* tables, const uint8_t* windows, uint32_t* out, int iters) {
    const uint32_t w = blockIdx.x, lane = threadIdx.x;
    uint32_t* tab = tables + (size_t)w * 4096;
    const uint8_t* win = windows + (size_t)w * 65536;
    uint32_t x = w * 2654435761u + lane * 40503u + 12345u, acc = 0;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const uint32_t h = (x >> 12) & 4095u;
        uint32_t old = 0;
        if (lane < 16) old = __hip_atomic_load(tab + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // table gather
        const uint32_t cand = (old ^ x) & 0xFFF0u;
        uint64_t b0 = 0, b1 = 0;
        if (lane < 16) { b0 = *(const uint64_t*)(win + cand); b1 = *(const uint64_t*)(win + cand + 8); }   // dependent candidate gather
        acc += (uint32_t)(b0 ^ (b1 >> 7));
        if (lane < 16) tab[h] = it + (uint32_t)b0;                                                         // commit
        x ^= acc;                                                                                          // next step depends on this one
    }
    if (acc == 0x12345) out[w] = acc;
}

int main(int argc, char** argv) {
    const int iters = 20000;
    for (int waves_per_cu : {10, 16, 24, 32}) {
        const int nw = 256 * waves_per_cu;
        uint32_t *tables, *out; uint8_t* windows;
        hipMalloc(&tables, (size_t)nw * 16384); hipMalloc(&windows, (size_t)nw * 65536); hipMalloc(&out, nw * 4);
        hipMemset(tables, 1, (size_t)nw * 16384); hipMemset(windows, 3, (size_t)nw * 65536);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(probe, dim3(nw), dim3(64), 0, 0, tables, windows, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(probe, dim3(nw), dim3(64), 0, 0, tables, windows, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("waves/CU %2d  resident waves %5d  table set %4.0f MB  window set %5.0f MB: %.2f G steps/s (%.0f cycles/step/wave at 2.1 GHz)\n",
               waves_per_cu, nw, nw * 16384.0 / 1e6, nw * 65536.0 / 1e6, (double)nw * iters / ms / 1e6, ms * 1e-3 * 2.1e9 / iters);
        hipFree(tables); hipFree(windows); hipFree(out);
    }
    return 0;
}
