// dist_gather.hip — liblzfear_dist.so: the frame reassembly of the block-sharded compressor (BASELINE configs[3]) over RCCL, behind the
// C ABI of include/lzfear_dist.h.  Replaces what the reference's single-threaded block loop does by construction — blocks written one
// after the other into one writer, src/framed/compress.rs:243-258,:277 — for blocks that were compressed on different GPUs.
// One process per GPU; xGMI is point-to-point, so the payload exchange is a grouped ncclSend / ncclRecv of exact-size segments (every
// link carries one segment once), not a padded ring all-gather.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
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

// Where librccl came from in this process (dladdr on the entry this library calls): bench.py prints it beside torch's own copy.
const char* lzf_dist_rccl_path(void) {
    static thread_local std::string path;
    Dl_info info;
    path = dladdr(reinterpret_cast<void*>(&ncclCommInitRank), &info) && info.dli_fname ? info.dli_fname : "";
    return path.c_str();
}

// The call is COLLECTIVE: whatever goes wrong on one rank before the payload exchange — an argument only this rank can judge
// (frame_cap, header, n_local), a failed launch, a block status that cannot be framed — travels to every rank in the size table
// (kRowExtra words per rank behind its size words: an error code and the rank's frame_cap), so that all ranks return the same
// failure BEFORE any of them posts a send or a receive.  (a rank that returned early left its peers in ncclRecv.)
int lzf_frame_gather(lzf_dist_comm* comm, const lzf_job_result* d_results, const uint8_t* d_comp, const uint8_t* d_src,
                     uint64_t stride, uint64_t block_size, uint32_t n_local, uint32_t n_blocks, uint64_t last_block_len,
                     const uint8_t* header, uint32_t header_len, uint8_t* d_frame, uint64_t frame_cap,
                     uint64_t* frame_len, uint64_t* comp_total, void* hip_stream) {
    // what every rank passes alike (a mismatch here is a bug of the caller on all ranks at once): safe to return on
    if (!comm || !frame_len || block_size == 0 || block_size > 0x7FFFFFFFull || last_block_len == 0 || last_block_len > block_size || n_blocks == 0)
        return fail(LZF_E_INVALID, "lzf_frame_gather", "bad argument");
    const int rank = comm->rank, world = comm->world;
    uint32_t lo, hi;
    shard_range(n_blocks, rank, world, lo, hi);
    // what only this rank can judge: carried to the others as local_err
    constexpr uint32_t kRowExtra = 4u;       // [0] error code, [1..2] frame_cap, [3] spare
    enum : uint32_t { kErrNone = 0u, kErrArgs = 1u, kErrShare = 2u, kErrHeader = 3u, kErrLaunch = 4u, kErrStatus = 5u };
    uint32_t local_err = kErrNone;
    if (!d_frame || !header || (n_local && (!d_results || !d_comp || !d_src)) || stride < block_size) local_err = kErrArgs;
    else if (hi - lo != n_local) local_err = kErrShare;
    // the frame this call writes has no block checksums and no content checksum (XXH32 does not compose across ranks): a header that
    // announces either would make the frame invalid (src/framed/compress.rs:259-263,:279-281; flags header.rs:8-16)
    else if (header_len < 7u || (header[4] & 0x14u) != 0u) local_err = kErrHeader;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const uint32_t max_blocks = (n_blocks + (uint32_t)world - 1u) / (uint32_t)world;
    const uint32_t row = max_blocks + kRowExtra;
    const uint32_t n_mine = local_err ? 0u : n_local;
    // scratch: [tab (row u32)][tabs (world x row u32)][bad u32][sp][dp][len]
    const size_t o_tabs = ((size_t)row * 4u + 255u) & ~(size_t)255u, o_bad = o_tabs + (((size_t)world * row * 4u + 255u) & ~(size_t)255u);
    const size_t o_sp = o_bad + 256u, o_dp = o_sp + (((size_t)n_local * 8u + 255u) & ~(size_t)255u), o_len = o_dp + (o_dp - o_sp);
    uint8_t* scratch = nullptr;
    HIPOK(hipMallocAsync(reinterpret_cast<void**>(&scratch), o_len + (o_dp - o_sp) + 256u, st));      // (without memory for the table this rank cannot tell anybody: the one early return left)
    struct Free { uint8_t* p; hipStream_t s; ~Free() { if (p) (void)hipFreeAsync(p, s); } } guard{scratch, st};
    uint32_t* tab = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* tabs = reinterpret_cast<uint32_t*>(scratch + o_tabs);
    uint32_t* bad = reinterpret_cast<uint32_t*>(scratch + o_bad);
    if (hipMemsetAsync(scratch, 0, o_sp, st) != hipSuccess) { (void)hipGetLastError(); local_err = local_err ? local_err : kErrLaunch; }
    if (n_mine) {
        hipLaunchKernelGGL(lzf_gather_sizes_kernel, dim3(1), dim3(1024), 0, st, d_results, n_mine, lo, n_blocks, block_size, last_block_len, tab, bad);
        if (hipGetLastError() != hipSuccess) local_err = kErrLaunch;
    }
    {   // this rank's error word and frame_cap behind its size words
        const uint32_t extra[kRowExtra] = {local_err, (uint32_t)frame_cap, (uint32_t)(frame_cap >> 32), 0u};
        if (hipMemcpyAsync(tab + max_blocks, extra, sizeof extra, hipMemcpyHostToDevice, st) != hipSuccess) { (void)hipGetLastError(); return fail(LZF_E_HIP, "lzf_frame_gather", "cannot publish this rank's state"); }
        if (hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); return fail(LZF_E_HIP, "lzf_frame_gather", "stream"); }      // (`extra` is on this stack frame)
    }
    // ---- the size table (one u32 per block, padded to the largest range, + the extra words)
    if (world > 1) NCCLOK(ncclAllGather(tab, tabs, row, ncclUint32, comm->comm, st));
    else HIPOK(hipMemcpyAsync(tabs, tab, (size_t)row * 4u, hipMemcpyDeviceToDevice, st));
    std::vector<uint32_t> h_tabs((size_t)world * row + 1u);
    HIPOK(hipMemcpyAsync(h_tabs.data(), tabs, (size_t)world * row * 4u, hipMemcpyDeviceToHost, st));
    HIPOK(hipMemcpyAsync(&h_tabs[(size_t)world * row], bad, 4u, hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));                                 // (segment sizes are arguments of the sends: the host needs them)
    // ---- every rank looks at every rank's state: from here on all ranks take the same way out
    uint32_t worst = 0; int worst_rank = -1; uint64_t min_cap = ~0ull;
    for (int r = 0; r < world; ++r) {
        const uint32_t* x = &h_tabs[(size_t)r * row + max_blocks];
        if (x[0] && !worst) { worst = x[0]; worst_rank = r; }
        const uint64_t cap_r = (uint64_t)x[1] | ((uint64_t)x[2] << 32);
        if (cap_r < min_cap) min_cap = cap_r;
    }
    if (worst) {
        char msg[160];
        snprintf(msg, sizeof msg, "rank %d: %s", worst_rank, worst == kErrArgs ? "bad argument" : worst == kErrShare ? "n_local is not its share of n_blocks"
                 : worst == kErrHeader ? "the header announces block or content checksums, which this frame does not carry" : "a launch failed");
        return fail(worst == kErrLaunch ? LZF_E_HIP : LZF_E_INVALID, "lzf_frame_gather", msg);
    }
    std::vector<uint64_t> seg_off((size_t)world + 1u);
    seg_off[0] = header_len;
    uint64_t payload = 0;
    // a rank with a status that cannot be framed says so through the size table: its word is zero-length (never a legal payload)
    uint32_t any_bad = h_tabs[(size_t)world * row];
    for (int r = 0; r < world; ++r) {
        uint32_t rlo, rhi;
        shard_range(n_blocks, r, world, rlo, rhi);
        uint64_t sg = 0;
        for (uint32_t i = 0; i < rhi - rlo; ++i) {
            const uint32_t n = h_tabs[(size_t)r * row + i] & 0x7FFFFFFFu;
            if (n == 0u) any_bad = 1u;
            sg += (uint64_t)n + 4u; payload += n;
        }
        seg_off[(size_t)r + 1u] = seg_off[r] + sg;
    }
    if (any_bad) return fail(LZF_CONTRACT, "lzf_frame_gather", "a block's compress status is neither OK nor OUTPUT_FULL (or its payload is empty)");
    const uint64_t flen = seg_off[world] + 4u;
    if (flen > min_cap) return fail(LZF_E_INVALID, "lzf_frame_gather", "frame_cap too small (on some rank)");
    // ---- this rank's segment, packed in place
    int pack_rc = LZF_OK;
    if (n_local) {
        const uint8_t** sp = reinterpret_cast<const uint8_t**>(scratch + o_sp);
        uint8_t** dp = reinterpret_cast<uint8_t**>(scratch + o_dp);
        uint64_t* len = reinterpret_cast<uint64_t*>(scratch + o_len);
        hipLaunchKernelGGL(lzf_gather_pack_kernel, dim3(1), dim3(1024), 0, st, tab, n_local, d_comp, d_src, stride, d_frame, seg_off[rank], sp, dp, len);
        if (hipGetLastError() != hipSuccess) pack_rc = LZF_E_HIP;
        else pack_rc = lzf_copy_ranges(sp, dp, len, n_local, block_size, st);
    }
    // ---- exchange: my segment to every peer, theirs into their places (one grouped operation).  A rank whose packing failed still
    //      takes part (its peers are about to post their receives: the bytes it sends are then not its blocks, and it says so below)
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
    if (pack_rc != LZF_OK) return fail(pack_rc, "lzf_frame_gather", "packing this rank's segment failed");
    HIPOK(hipMemcpyAsync(d_frame, header, header_len, hipMemcpyHostToDevice, st));
    HIPOK(hipMemsetAsync(d_frame + flen - 4u, 0, 4u, st));           // EndMark (compress.rs:277)
    HIPOK(hipStreamSynchronize(st));
    *frame_len = flen;
    if (comp_total) *comp_total = payload;
    return LZF_OK;
}

}  // extern "C"
