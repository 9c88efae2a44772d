// lz4_compress_team.hip — the latency class of lzf_compress_batch for gfx950: one block per compute unit, a team of three wavefronts
// (searcher, emitter, feeder), the input window and the position table in LDS.  The kernel body is lz4_compress_team.inc (written
// against lzf_simt.h so that the CPU suite runs the same source under a lock-step emulator); this file instantiates it with the
// gfx950 primitives.  160 KiB of LDS per workgroup (all of a CU's): one workgroup per CU.
#include "lz4_compress_team.inc"
#include "kernels.h"

namespace lzf {

static_assert(team::kLdsWords * 4u == 163840u, "capi.hip's kTeamLds (the dispatch's LDS requirement) is this number");

__global__ __launch_bounds__(192) void lzf_compress_team_kernel(const lzf_compress_job* __restrict__ jobs, lzf_job_result* __restrict__ results,
                                                                uint32_t n_jobs, const uint32_t* __restrict__ perm, uint32_t alone) {
    // aligned(16384): hash5_slot_addr (lzf_simt.h) ORs a slot's offset into the table's address, which is only an addition while the
    // table (at byte 0x24000 of this array) starts on a 16 KiB boundary; a second __shared__ object in front of this one would break
    // that silently — hence the alignment AND the check (the emulator adds, so the CPU suite cannot see it)
    __shared__ __attribute__((aligned(16384))) uint32_t lds[team::kLdsWords];
    if ((lds_addr(lds) & 0x3FFFu) != 0u) __builtin_trap();
    const SimtGpu b{lds, lds_addr(lds)};
    const team::Args a{jobs, results, n_jobs, perm, alone};
    team::compress_team<SimtGpu, false>(b, a, blockIdx.x);
}

// the jobs the kernel above leaves to lzf_compress_wave_kernel but a team can do: caller-owned U32 tables at any offset (linked streams)
__global__ __launch_bounds__(192) void lzf_compress_team_carry_kernel(const lzf_compress_job* __restrict__ jobs, lzf_job_result* __restrict__ results,
                                                                      uint32_t n_jobs, const uint32_t* __restrict__ perm) {
    __shared__ __attribute__((aligned(16384))) uint32_t lds[team::kLdsWords];
    if ((lds_addr(lds) & 0x3FFFu) != 0u) __builtin_trap();
    const SimtGpu b{lds, lds_addr(lds)};
    const team::Args a{jobs, results, n_jobs, perm, 0u};
    team::compress_team<SimtGpu, true>(b, a, blockIdx.x);
}

}  // namespace lzf
