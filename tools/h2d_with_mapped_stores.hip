// h2d_with_mapped_stores.hip — does the host link carry both directions at once when the way back is KERNEL STORES into host-mapped
// pinned memory instead of a copy engine / blit kernel?  (round-4 review item 5: lzf_frame_decompress_many's upload and download
// measured 10.0 + 9.4 ms alone and 18.8 ms together with hipMemcpyAsync both ways, profiles/r03_e2e_overlap_findings.txt.)
//   a  H2D alone        hipMemcpyAsync pinned -> device
//   b  stores alone     a kernel reads device memory and writes 16 bytes per lane to hipHostMalloc(mapped) memory
//   c  a + b together   two streams
//   d  H2D + D2H        hipMemcpyAsync both ways on two streams (the round-3 observation, for the same box)
//   e  D2H alone        hipMemcpyAsync device -> pinned
// build: hipcc --offload-arch=gfx950 -O3 -o tools/h2d_with_mapped_stores tools/h2d_with_mapped_stores.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void store_to_host(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t N = (argc > 1 ? (size_t)atol(argv[1]) : 1024) << 20;
    const int blocks = argc > 2 ? atoi(argv[2]) : 512;
    void *hA, *hB, *hM, *dD, *dS, *dM;
    OK(hipHostMalloc(&hA, N, hipHostMallocDefault)); OK(hipHostMalloc(&hB, N, hipHostMallocDefault));
    OK(hipHostMalloc(&hM, N, hipHostMallocMapped)); OK(hipHostGetDevicePointer(&dM, hM, 0));
    OK(hipMalloc(&dD, N)); OK(hipMalloc(&dS, N));
    memset(hA, 1, N); memset(hB, 0, N); memset(hM, 0, N); OK(hipMemset(dS, 7, N));
    hipStream_t s1, s2; OK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); OK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto run = [&](const char* name, bool h2d, int back /* 0 none, 1 mapped stores, 2 memcpy D2H */) -> int {
        double best = 1e9;
        for (int it = 0; it < 4; ++it) {
            OK(hipDeviceSynchronize());
            const double t0 = now();
            if (h2d) OK(hipMemcpyAsync(dD, hA, N, hipMemcpyHostToDevice, s1));
            if (back == 1) hipLaunchKernelGGL(store_to_host, dim3(blocks), dim3(256), 0, s2, (const u32x4*)dS, (u32x4*)dM, N / 16);
            if (back == 2) OK(hipMemcpyAsync(hB, dS, N, hipMemcpyDeviceToHost, s2));
            OK(hipStreamSynchronize(s1)); OK(hipStreamSynchronize(s2));
            const double dt = now() - t0;
            if (it && dt < best) best = dt;
        }
        printf("%-34s %7.2f ms  (%.1f GB/s per direction)\n", name, best * 1e3, (double)N / best / 1e9);
        return 0;
    };
    printf("%zu MiB per direction, store kernel: %d workgroups of 256\n", N >> 20, blocks);
    if (run("a  H2D alone (hipMemcpyAsync)", true, 0)) return 1;
    if (run("e  D2H alone (hipMemcpyAsync)", false, 2)) return 1;
    if (run("b  mapped-host stores alone", false, 1)) return 1;
    if (run("c  H2D + mapped-host stores", true, 1)) return 1;
    if (run("d  H2D + D2H (hipMemcpyAsync)", true, 2)) return 1;
    // the stores arrived?
    const unsigned char* m = (const unsigned char*)hM; size_t bad = 0; for (size_t i = 0; i < N; i += 4097) bad += m[i] != 7;
    printf("mapped buffer check: %zu mismatches\n", bad);
    return 0;
}
