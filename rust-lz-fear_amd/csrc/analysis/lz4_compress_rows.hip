// lz4_compress_rows.hip — the row-mapped compress kernel for gfx950: four blocks per wavefront, persistent waves, a job queue.
// The kernel body is lz4_compress_rows.inc (written against lzf_simt.h so that the CPU suite can run the same source under a
// lock-step emulator); this file instantiates it with the gfx950 primitives.  One wave per workgroup, 34 KiB of LDS: four
// workgroups per CU, one per SIMD.
#include "lz4_compress_rows.inc"
#include "../kernels.h"

namespace lzf {

__global__ __launch_bounds__(64) void lzf_compress_rows_kernel(const lzf_compress_job* __restrict__ jobs, lzf_job_result* __restrict__ results,
                                                               uint32_t n_jobs, const uint32_t* __restrict__ perm, uint32_t* __restrict__ queue,
                                                               uint32_t rows_active, uint32_t alone) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[rows::kLdsWords];
    const SimtGpu b{lds, lds_addr(lds)};
    const rows::Args a{jobs, results, n_jobs, perm, queue, rows_active, alone};
    rows::compress_rows_wave(b, a);
}

}  // namespace lzf
