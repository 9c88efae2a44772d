// Micro-test (gfx950): are byte-misaligned LDS accesses correct when issued directly
// (inline asm ds_read/ds_write b16/b32/b64), and what do they cost?  hipcc itself never emits
// them (it splits packed accesses into bytes), so this tells whether hand-issued ones are usable.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

template <int W> __device__ __forceinline__ void copy_w(uint32_t src, uint32_t dst);
template <> __device__ __forceinline__ void copy_w<1>(uint32_t src, uint32_t dst) {
    uint32_t v; asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b8 %2, %0" : "=&v"(v) : "v"(src), "v"(dst) : "memory");
}
template <> __device__ __forceinline__ void copy_w<2>(uint32_t src, uint32_t dst) {
    uint32_t v; asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b16 %2, %0" : "=&v"(v) : "v"(src), "v"(dst) : "memory");
}
template <> __device__ __forceinline__ void copy_w<4>(uint32_t src, uint32_t dst) {
    uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b32 %2, %0" : "=&v"(v) : "v"(src), "v"(dst) : "memory");
}
template <> __device__ __forceinline__ void copy_w<8>(uint32_t src, uint32_t dst) {
    uint64_t v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b64 %2, %0" : "=&v"(v) : "v"(src), "v"(dst) : "memory");
}

template <int W, int MIS, int STRIDE>
__global__ void k(uint8_t* gout, uint32_t* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 64 + 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 64 + 64; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)lds;   // LDS byte address
    const uint32_t src = base + lane * STRIDE + MIS, dst = base + lane * STRIDE + 32 + MIS;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) copy_w<W>(src, dst);
    long long t1 = clock64();
    __syncthreads();
    for (int i = lane; i < 64 * 64; i += 64) gout[i] = lds[i];
    if (lane == 0) cyc[0] = (uint32_t)(t1 - t0);
}

template <int W, int MIS, int STRIDE>
void run() {
    uint8_t* d; uint32_t* c;
    hipMalloc(&d, 64 * 64); hipMalloc(&c, 4);
    const int iters = 1000;
    hipLaunchKernelGGL((k<W, MIS, STRIDE>), dim3(1), dim3(64), 0, 0, d, c, iters);
    hipDeviceSynchronize();
    std::vector<uint8_t> h(64 * 64); uint32_t cy;
    hipMemcpy(h.data(), d, h.size(), hipMemcpyDeviceToHost); hipMemcpy(&cy, c, 4, hipMemcpyDeviceToHost);
    std::vector<uint8_t> e(64 * 64 + 64);
    for (size_t i = 0; i < e.size(); ++i) e[i] = (uint8_t)(i * 7 + 3);
    for (int lane = 0; lane < 64; ++lane) memcpy(&e[lane * STRIDE + 32 + MIS], &e[lane * STRIDE + MIS], W);
    bool ok = memcmp(e.data(), h.data(), 64 * 64) == 0;
    printf("b%-3d misalign=%d stride=%2d  %s  %.1f cycles per read+wait+write (64 lanes)\n", W * 8, MIS, STRIDE, ok ? "OK   " : "WRONG", cy / (double)iters);
    hipFree(d); hipFree(c);
}

int main() {
    run<1, 0, 64>(); run<1, 1, 64>(); run<1, 1, 61>();
    run<2, 0, 64>(); run<2, 1, 64>(); run<2, 1, 61>();
    run<4, 0, 64>(); run<4, 1, 64>(); run<4, 2, 64>(); run<4, 3, 64>(); run<4, 1, 61>(); run<4, 0, 60>();
    run<8, 0, 64>(); run<8, 1, 64>(); run<8, 4, 64>(); run<8, 3, 61>();
    return 0;
}
