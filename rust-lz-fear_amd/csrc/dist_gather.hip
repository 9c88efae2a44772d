// dist_gather.hip — liblzfear_dist.so: the frame reassembly of the block-sharded compressor (BASELINE configs[3]) over RCCL, behind the
// C ABI of include/lzfear_dist.h.  Replaces what the reference's single-threaded block loop does by construction — blocks written one
// after the other into one writer, src/framed/compress.rs:243-258,:277 — for blocks that were compressed on different GPUs.
// One process per GPU; xGMI is point-to-point, so the payload exchange is a grouped ncclSend / ncclRecv of exact-size segments (every
// link carries one segment once), not a padded ring all-gather.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/lzfear_dist.h"

struct lzf_dist_comm { ncclComm_t comm; int rank, world; };

namespace {
thread_local std::string g_err;
int fail(int code, const char* what, const char* detail) { g_err = std::string(what) + ": " + detail; return code; }
#define HIPOK(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return fail(LZF_E_HIP, #expr, hipGetErrorString(e__)); } while (0)
#define NCCLOK(expr) do { ncclResult_t r__ = (expr); if (r__ != ncclSuccess) return fail(LZF_E_HIP, #expr, ncclGetErrorString(r__)); } while (0)

// blocks [lo, hi) of rank r: contiguous ranges, the first n % world ranks one block more (dist.shard_range)
inline void shard_range(uint32_t n, int r, int world, uint32_t& lo, uint32_t& hi) {
    const uint32_t base = n / (uint32_t)world, extra = n % (uint32_t)world;
    lo = (uint32_t)r * base + ((uint32_t)r < extra ? (uint32_t)r : extra);
    hi = lo + base + ((uint32_t)r < extra ? 1u : 0u);
}

// One workgroup: payload length and size word of every local block (compress.rs:244-258), bad statuses flagged.
__global__ __launch_bounds__(1024) void lzf_gather_sizes_kernel(const lzf_job_result* __restrict__ res, uint32_t n_local, uint32_t first_block, uint32_t n_blocks,
                                                               uint64_t block_size, uint64_t last_block_len, uint32_t* __restrict__ tab /* max_blocks, zero-filled */,
                                                               uint32_t* __restrict__ bad) {
    for (uint32_t i = threadIdx.x; i < n_local; i += blockDim.x) {
        const uint64_t raw = first_block + i + 1u == n_blocks ? last_block_len : block_size;
        const int st = res[i].status;
        if (st != LZF_OK && st != LZF_OUTPUT_FULL) atomicOr(bad, 1u);
        // the size word itself (stored bit = the raw block travels); a status that cannot be framed leaves 0 — no legal payload is empty
        tab[i] = st == LZF_OK ? (uint32_t)res[i].out_len : st == LZF_OUTPUT_FULL ? ((uint32_t)raw | 0x80000000u) : 0u;
    }
}
// One workgroup: where every local block goes in the frame (exclusive scan of payload + 4), its size word written, the copy lists filled.
__global__ __launch_bounds__(1024) void lzf_gather_pack_kernel(const uint32_t* __restrict__ tab, uint32_t n_local, const uint8_t* __restrict__ comp, const uint8_t* __restrict__ src,
                                                              uint64_t stride, uint8_t* __restrict__ frame, uint64_t seg_base,
                                                              const uint8_t** __restrict__ sp, uint8_t** __restrict__ dp, uint64_t* __restrict__ len) {
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x, per = (n_local + 1023u) / 1024u;
    const uint32_t a = t * per < n_local ? t * per : n_local, b = a + per < n_local ? a + per : n_local;
    uint64_t s = 0;
    for (uint32_t i = a; i < b; ++i) s += (uint64_t)(tab[i] & 0x7FFFFFFFu) + 4u;
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {                      // inclusive scan
        const uint64_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t off = seg_base + part[t] - s;
    for (uint32_t i = a; i < b; ++i) {
        const uint32_t w = tab[i], n = w & 0x7FFFFFFFu;
        frame[off] = (uint8_t)w; frame[off + 1] = (uint8_t)(w >> 8); frame[off + 2] = (uint8_t)(w >> 16); frame[off + 3] = (uint8_t)(w >> 24);
        sp[i] = ((w >> 31) ? src : comp) + (uint64_t)i * stride;     // stored: the raw block (compress.rs:250-255)
        dp[i] = frame + off + 4;
        len[i] = n;
        off += (uint64_t)n + 4u;
    }
}
}  // namespace

extern "C" {

const char* lzf_dist_last_error(void) { return g_err.c_str(); }

int lzf_dist_unique_id(uint8_t id[LZF_DIST_UNIQUE_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == LZF_DIST_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id) return fail(LZF_E_INVALID, "lzf_dist_unique_id", "NULL");
    ncclUniqueId u;
    NCCLOK(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return LZF_OK;
}

int lzf_dist_comm_init(const uint8_t id[LZF_DIST_UNIQUE_ID_BYTES], int rank, int world, lzf_dist_comm** comm) {
    if (!id || !comm || world < 1 || rank < 0 || rank >= world) return fail(LZF_E_INVALID, "lzf_dist_comm_init", "bad argument");
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t c = nullptr;
    NCCLOK(ncclCommInitRank(&c, world, u, rank));
    *comm = new lzf_dist_comm{c, rank, world};
    return LZF_OK;
}

int lzf_dist_comm_count(const lzf_dist_comm* comm) {
    if (!comm) return 0;
    int n = 0;
    if (ncclCommCount(comm->comm, &n) != ncclSuccess) return 0;
    return n;
}

void lzf_dist_comm_free(lzf_dist_comm* comm) {
    if (!comm) return;
    (void)ncclCommDestroy(comm->comm);
    delete comm;
}

int lzf_frame_gather(lzf_dist_comm* comm, const lzf_job_result* d_results, const uint8_t* d_comp, const uint8_t* d_src,
                     uint64_t stride, uint64_t block_size, uint32_t n_local, uint32_t n_blocks, uint64_t last_block_len,
                     const uint8_t* header, uint32_t header_len, uint8_t* d_frame, uint64_t frame_cap,
                     uint64_t* frame_len, uint64_t* comp_total, void* hip_stream) {
    if (!comm || !d_frame || !header || !frame_len || (n_local && (!d_results || !d_comp || !d_src)) || block_size == 0 || block_size > 0x7FFFFFFFull ||
        stride < block_size || last_block_len == 0 || last_block_len > block_size || n_blocks == 0)
        return fail(LZF_E_INVALID, "lzf_frame_gather", "bad argument");
    const int rank = comm->rank, world = comm->world;
    uint32_t lo, hi;
    shard_range(n_blocks, rank, world, lo, hi);
    if (hi - lo != n_local) return fail(LZF_E_INVALID, "lzf_frame_gather", "n_local is not this rank's share of n_blocks");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const uint32_t max_blocks = (n_blocks + (uint32_t)world - 1u) / (uint32_t)world;
    // scratch: [tab (max_blocks u32)][tabs (world x max_blocks u32)][bad u32][sp][dp][len]
    const size_t o_tabs = ((size_t)max_blocks * 4u + 255u) & ~(size_t)255u, o_bad = o_tabs + (((size_t)world * max_blocks * 4u + 255u) & ~(size_t)255u);
    const size_t o_sp = o_bad + 256u, o_dp = o_sp + (((size_t)n_local * 8u + 255u) & ~(size_t)255u), o_len = o_dp + (o_dp - o_sp);
    uint8_t* scratch = nullptr;
    HIPOK(hipMallocAsync(reinterpret_cast<void**>(&scratch), o_len + (o_dp - o_sp) + 256u, st));
    struct Free { uint8_t* p; hipStream_t s; ~Free() { if (p) (void)hipFreeAsync(p, s); } } guard{scratch, st};
    uint32_t* tab = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* tabs = reinterpret_cast<uint32_t*>(scratch + o_tabs);
    uint32_t* bad = reinterpret_cast<uint32_t*>(scratch + o_bad);
    HIPOK(hipMemsetAsync(scratch, 0, o_sp, st));
    if (n_local) hipLaunchKernelGGL(lzf_gather_sizes_kernel, dim3(1), dim3(1024), 0, st, d_results, n_local, lo, n_blocks, block_size, last_block_len, tab, bad);
    HIPOK(hipGetLastError());
    // ---- the size table (one u32 per block, padded to the largest range)
    if (world > 1) NCCLOK(ncclAllGather(tab, tabs, max_blocks, ncclUint32, comm->comm, st));
    else HIPOK(hipMemcpyAsync(tabs, tab, (size_t)max_blocks * 4u, hipMemcpyDeviceToDevice, st));
    std::vector<uint32_t> h_tabs((size_t)world * max_blocks + 1u);
    HIPOK(hipMemcpyAsync(h_tabs.data(), tabs, (size_t)world * max_blocks * 4u, hipMemcpyDeviceToHost, st));
    HIPOK(hipMemcpyAsync(&h_tabs[(size_t)world * max_blocks], bad, 4u, hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));                                 // (segment sizes are arguments of the sends: the host needs them)
    uint32_t any_bad = h_tabs[(size_t)world * max_blocks];
    std::vector<uint64_t> seg_off((size_t)world + 1u);
    seg_off[0] = header_len;
    uint64_t payload = 0;
    for (int r = 0; r < world; ++r) {
        uint32_t rlo, rhi;
        shard_range(n_blocks, r, world, rlo, rhi);
        uint64_t s = 0;
        for (uint32_t i = 0; i < rhi - rlo; ++i) { const uint32_t n = h_tabs[(size_t)r * max_blocks + i] & 0x7FFFFFFFu; s += (uint64_t)n + 4u; payload += n; }
        seg_off[(size_t)r + 1u] = seg_off[r] + s;
    }
    // a rank with a status that cannot be framed says so to everybody through the size table: its words are zero-length (never a
    // legal payload) — every rank fails the same way instead of one rank leaving the others in a send
    if (world > 1) {
        for (int r = 0; r < world; ++r) {
            uint32_t rlo, rhi; shard_range(n_blocks, r, world, rlo, rhi);
            for (uint32_t i = 0; i < rhi - rlo; ++i) if ((h_tabs[(size_t)r * max_blocks + i] & 0x7FFFFFFFu) == 0u) any_bad = 1u;
        }
    }
    if (any_bad) return fail(LZF_CONTRACT, "lzf_frame_gather", "a block's compress status is neither OK nor OUTPUT_FULL (or its payload is empty)");
    const uint64_t flen = seg_off[world] + 4u;
    if (flen > frame_cap) return fail(LZF_E_INVALID, "lzf_frame_gather", "frame_cap too small");
    // ---- this rank's segment, packed in place
    if (n_local) {
        const uint8_t** sp = reinterpret_cast<const uint8_t**>(scratch + o_sp);
        uint8_t** dp = reinterpret_cast<uint8_t**>(scratch + o_dp);
        uint64_t* len = reinterpret_cast<uint64_t*>(scratch + o_len);
        hipLaunchKernelGGL(lzf_gather_pack_kernel, dim3(1), dim3(1024), 0, st, tab, n_local, d_comp, d_src, stride, d_frame, seg_off[rank], sp, dp, len);
        HIPOK(hipGetLastError());
        const int rc = lzf_copy_ranges(sp, dp, len, n_local, block_size, st);
        if (rc != LZF_OK) return fail(rc, "lzf_copy_ranges", lzf_last_error());
    }
    // ---- exchange: my segment to every peer, theirs into their places (one grouped operation)
    if (world > 1) {
        NCCLOK(ncclGroupStart());
        ncclResult_t gr = ncclSuccess;                                   // (a failed call inside the group still closes it)
        for (int p = 0; p < world && gr == ncclSuccess; ++p) {
            if (p == rank) continue;
            const uint64_t mine = seg_off[(size_t)rank + 1u] - seg_off[rank], theirs = seg_off[(size_t)p + 1u] - seg_off[p];
            if (mine) gr = ncclSend(d_frame + seg_off[rank], mine, ncclUint8, p, comm->comm, st);
            if (theirs && gr == ncclSuccess) gr = ncclRecv(d_frame + seg_off[p], theirs, ncclUint8, p, comm->comm, st);
        }
        const ncclResult_t ge = ncclGroupEnd();
        if (gr != ncclSuccess) return fail(LZF_E_HIP, "ncclSend / ncclRecv", ncclGetErrorString(gr));
        if (ge != ncclSuccess) return fail(LZF_E_HIP, "ncclGroupEnd", ncclGetErrorString(ge));
    }
    HIPOK(hipMemcpyAsync(d_frame, header, header_len, hipMemcpyHostToDevice, st));
    HIPOK(hipMemsetAsync(d_frame + flen - 4u, 0, 4u, st));           // EndMark (compress.rs:277)
    HIPOK(hipStreamSynchronize(st));
    *frame_len = flen;
    if (comp_total) *comp_total = payload;
    return LZF_OK;
}

}  // extern "C"
