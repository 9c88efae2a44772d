// aux_kernels.hip — small kernels around the codec: XXH32 of many buffers, dictionary
// seeding of a U32Table, EncoderTable::offset.
#include "lzf_device.h"
#include "kernels.h"

namespace lzf {

// ---------------------------------------------------------------------------------------
// XXH32 (seed 0) — the block checksums of src/framed/compress.rs:259-263 and
// src/framed/decompress.rs:228-235 (twox-hash XxHash32).  The four accumulators of one hash
// are four adjacent lanes; a wave hashes 16 buffers at a time.  Each accumulator is a serial
// multiply-rotate chain over its stripe words, so parallelism is 4 lanes x #buffers.
// ---------------------------------------------------------------------------------------
constexpr uint32_t XP1 = 2654435761u, XP2 = 2246822519u, XP3 = 3266489917u, XP4 = 668265263u, XP5 = 374761393u;
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t xround(uint32_t acc, uint32_t w) { return rotl32(acc + w * XP2, 13) * XP1; }

__global__ __launch_bounds__(64) void lzf_xxh32_kernel(const uint8_t* const* __restrict__ ptrs,
                                                       const uint64_t* __restrict__ lens,
                                                       uint32_t* __restrict__ out, uint32_t n) {
    const uint32_t g = (blockIdx.x * 64u + threadIdx.x) >> 2;
    const uint32_t q = threadIdx.x & 3u;
    const bool act = g < n;
    cgu8* p = act ? as_global(ptrs[g]) : nullptr;
    const uint64_t len = act ? lens[g] : 0;
    uint32_t v = q == 0 ? XP1 + XP2 : q == 1 ? XP2 : q == 2 ? 0u : 0u - XP1;
    const uint64_t stripes = len >> 4;
    cgu8* sp = p + q * 4u;
    uint64_t s = 0;
    for (; s + 4 <= stripes; s += 4) {   // 4 loads in flight per lane
        const uint32_t w0 = ld4(sp + (s + 0) * 16), w1 = ld4(sp + (s + 1) * 16);
        const uint32_t w2 = ld4(sp + (s + 2) * 16), w3 = ld4(sp + (s + 3) * 16);
        v = xround(xround(xround(xround(v, w0), w1), w2), w3);
    }
    for (; s < stripes; ++s) v = xround(v, ld4(sp + s * 16));
    // merge the four accumulators (lanes 4g..4g+3)
    const uint32_t r = q == 0 ? rotl32(v, 1) : q == 1 ? rotl32(v, 7) : q == 2 ? rotl32(v, 12) : rotl32(v, 18);
    uint32_t h = r + __shfl_xor(r, 1);
    h = h + __shfl_xor(h, 2);
    if (act && q == 0) {
        if (len < 16) h = XP5;   // seed + PRIME5
        h += (uint32_t)len;
        cgu8* t = p + (stripes << 4);
        uint32_t rem = (uint32_t)(len & 15u);
        while (rem >= 4) { h = rotl32(h + ld4(t) * XP3, 17) * XP4; t += 4; rem -= 4; }
        while (rem) { h = rotl32(h + (uint32_t)(*t) * XP5, 11) * XP1; ++t; --rem; }
        h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
        out[g] = h;
    }
}

// The same hash with one wavefront per buffer, for long buffers (content checksums, the checksums of 4 MiB blocks): the kernel
// above keeps only 64 bytes per hash in flight, which is dependent-load latency all the way.  Here the wave fetches 1 KiB per
// step with coalesced 16-byte loads, two steps ahead, multiplies by PRIME2 in all 64 lanes at once and parks the products in
// LDS; lanes 0..3 then run the serial chain (add, rotate, multiply — ~30 cycles a stripe, ~1.3 GB/s per hash) on LDS reads
// that do not depend on the chain.  The chip runs thousands of such chains at once.
__global__ __launch_bounds__(64) void lzf_xxh32_wave_kernel(const uint8_t* const* __restrict__ ptrs,
                                                            const uint64_t* __restrict__ lens,
                                                            uint32_t* __restrict__ out, uint32_t n) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[2][256];
    const uint32_t g = blockIdx.x;
    if (g >= n) return;
    const uint32_t lane = threadIdx.x, q = lane & 3u;
    cgu8* p = as_global(ptrs[g]);
    const uint64_t len = lens[g];
    uint32_t v = q == 0 ? XP1 + XP2 : q == 1 ? XP2 : q == 2 ? 0u : 0u - XP1;
    const uint64_t chunks = len >> 10;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    u32x4 r0 = chunks > 0 ? ld16(p + lane * 16u) : zero;
    u32x4 r1 = chunks > 1 ? ld16(p + 1024u + lane * 16u) : zero;
    for (uint64_t c = 0; c < chunks; ++c) {
        uint32_t* b = buf[c & 1u];
        *reinterpret_cast<u32x4*>(b + lane * 4u) = r0 * XP2;
        r0 = r1;
        r1 = c + 2 < chunks ? ld16(p + (c + 2) * 1024u + lane * 16u) : zero;
        __syncthreads();                               // (one wave: orders the LDS writes before the reads for the compiler)
#pragma unroll 16
        for (uint32_t s = 0; s < 64u; ++s) v = rotl32(v + b[s * 4u + q], 13) * XP1;
    }
    const uint64_t stripes = len >> 4;
    cgu8* sp = p + q * 4u;
    for (uint64_t s = chunks << 6; s < stripes; ++s) v = xround(v, ld4(sp + s * 16));
    const uint32_t r = q == 0 ? rotl32(v, 1) : q == 1 ? rotl32(v, 7) : q == 2 ? rotl32(v, 12) : rotl32(v, 18);
    uint32_t h = r + __shfl_xor(r, 1);
    h = h + __shfl_xor(h, 2);
    if (lane == 0) {
        if (len < 16) h = XP5;   // seed + PRIME5
        h += (uint32_t)len;
        cgu8* t = p + (stripes << 4);
        uint32_t rem = (uint32_t)(len & 15u);
        while (rem >= 4) { h = rotl32(h + ld4(t) * XP3, 17) * XP4; t += 4; rem -= 4; }
        while (rem) { h = rotl32(h + (uint32_t)(*t) * XP5, 11) * XP1; ++t; --rem; }
        h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
        out[g] = h;
    }
}

// ---------------------------------------------------------------------------------------
// Template-table seeding — src/framed/compress.rs:202-211:
//   for window in dict.windows(8).step_by(3) { template_table.replace(dict, offset) }
// on a default table.  Sequential semantics = "the last position with a given hash wins", and
// positions only grow, so the result is a per-slot maximum: order-free atomicMax.
// (Slot value 0 means both "empty" and "position 0", exactly as in the reference, mod.rs:34.)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lzf_seed_table_kernel(lzf_u32_table* __restrict__ t,
                                                             const uint8_t* __restrict__ dict, uint64_t dict_len) {
    if (dict_len < 8) return;
    const uint64_t count = (dict_len - 8) / 3 + 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t o = i * 3;
        const uint64_t v8 = ld8(as_global(dict) + o);
        const uint32_t h = (uint32_t)(((v8 << 24) * 889523592379ull) >> 52);
        atomicMax(&t->dict[h], (uint32_t)o);
    }
}

// EncoderTable::offset — src/raw/compress/mod.rs:72-74 (U32Table), :97-99 (U16Table)
__global__ void lzf_table_offset_kernel(void* table, uint32_t kind, uint64_t add) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (kind == LZF_TABLE_U32) ((lzf_u32_table*)table)->offset += add;
        else ((lzf_u16_table*)table)->offset += add;
    }
}

__global__ void lzf_table_offset_batch_kernel(void* const* __restrict__ tables, const uint64_t* __restrict__ adds, uint32_t n, uint32_t kind) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !adds[i]) return;
    if (kind == LZF_TABLE_U32) ((lzf_u32_table*)tables[i])->offset += adds[i];
    else ((lzf_u16_table*)tables[i])->offset += adds[i];
}

// One workgroup per linked-block stream, between two decode steps (lzfear_hip.h: lzf_chain_decompress_step).
__global__ __launch_bounds__(256) void lzf_chain_decompress_step_kernel(const lzf_chain_step* __restrict__ steps, lzf_chain_state* __restrict__ state,
                                                                        uint32_t n, lzf_decompress_job* __restrict__ jobs,
                                                                        const lzf_job_result* __restrict__ results) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const lzf_chain_step st = steps[i];
    __shared__ lzf_chain_state cs_in;                              // read once: lane 0 stores the new state below while other waves may not have started
    if (threadIdx.x == 0) cs_in = state[i];
    __syncthreads();
    lzf_chain_state cs = cs_in;
    if (st.prev_job != 0xFFFFFFFFu && !cs.dead) {                 // finish the previous step (decompress.rs:253-269: the output joins the history)
        const lzf_job_result r = results[st.prev_job];
        if (r.status != LZF_OK) cs.dead = 1u;
        else {
            if (r.out_len - cs.length > st.block_maxsize) cs.dead = 1u;      // decompress.rs:272-274 BlockSizeOverflow ends the stream
            cs.length = r.out_len;
        }
    }
    if (st.job != 0xFFFFFFFFu) {
        if (threadIdx.x == 0) {
            lzf_decompress_job& j = jobs[st.job];
            if (cs.dead) { j.input_len = 0; j.out_existing_len = 0; j.out_cap = 0; j.output_limit = 0; }
            else { j.out_existing_len = cs.length; j.out_cap = cs.length + st.block_maxsize + j.input_len; j.output_limit = cs.length + st.block_maxsize; }
        }
    } else if (st.stored_len && !cs.dead) {                       // decompress.rs:250: a stored block is appended as it is
        cgu8* s = as_global(st.stored_src);
        gu8* d = as_global(st.out) + cs.length;
        for (uint64_t t = threadIdx.x; t < st.stored_len; t += blockDim.x) d[t] = s[t];
        cs.length += st.stored_len;
    }
    if (threadIdx.x == 0) state[i] = cs;
}

// Copies ranges [src[r], src[r] + len[r]) -> dst[r]: blockIdx.y = range, blockIdx.x = 64 KiB piece of it,
// 256 threads x 16 bytes per step (unaligned 16-byte accesses are fine on gfx950).
__global__ __launch_bounds__(256) void lzf_copy_ranges_kernel(const uint8_t* const* __restrict__ src, uint8_t* const* __restrict__ dst,
                                                              const uint64_t* __restrict__ len, uint32_t n) {
    const uint32_t r = blockIdx.y;
    if (r >= n) return;
    const uint64_t total = len[r];
    const uint64_t a = (uint64_t)blockIdx.x * 65536ull;
    if (a >= total) return;
    const uint64_t b = a + 65536ull < total ? a + 65536ull : total;
    cgu8* s = as_global(src[r]);
    gu8* d = as_global(dst[r]);
    uint64_t i = a + (uint64_t)threadIdx.x * 16ull;
    for (; i + 16ull <= b; i += 256ull * 16ull) st16(d + i, ld16(s + i));
    if (i < b) for (uint64_t t = i; t < b; ++t) d[t] = s[t];     // the piece's last, partial 16 bytes (one thread)
}

// ---- job ordering for lzf_compress_batch (capi.hip) -------------------------------------------------------------------
// A compress job's cost is not known from its size (a 4 MiB block takes 0.1 .. 0.5 s of one wavefront, depending on the
// data), and with a few rounds of one-wave jobs the launch finishes when the last long job does.  So the batch is probed
// first: kCostSample bytes from the middle of every large payload are compressed with the output dropped
// (lzf_compress_compact_kernel<true>), and the real kernels then take the jobs longest first.
__device__ __forceinline__ uint64_t cost_payload(const lzf_compress_job& j) { return j.cursor < j.input_len ? j.input_len - j.cursor : 0ull; }
// a job is probed with `parts` pieces of `piece` bytes spread evenly over its payload (only payloads of >= 4 x that much)
__device__ __forceinline__ uint32_t cost_sample_len(uint64_t payload, uint32_t piece, uint32_t parts) {
    return payload >= 4ull * piece * parts ? piece * parts : 0u;
}

__global__ __launch_bounds__(256) void lzf_cost_probe_jobs_kernel(const lzf_compress_job* __restrict__ jobs,
                                                                  lzf_compress_job* __restrict__ probes, uint32_t n,
                                                                  uint32_t piece, uint32_t parts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;     // probe i = piece (i % parts) of job (i / parts)
    if (i >= n * parts) return;
    const lzf_compress_job j = jobs[i / parts];
    const uint64_t payload = cost_payload(j);
    const uint32_t sl = cost_sample_len(payload, piece, parts);
    const uint32_t k = i % parts;
    lzf_compress_job p;
    p.input = j.input + j.cursor + (sl ? (((payload - piece) * (2u * k + 1u) / (2u * parts)) & ~15ull) : 0ull);
    p.input_len = sl ? piece : 0u;    // 0: small job, not probed (its estimate is its size)
    p.cursor = 0;
    p.out = nullptr;                  // the probe stores nothing
    p.out_cap = ~0ull;
    p.table = nullptr;
    p.table_kind = LZF_TABLE_U32;
    p.flags = 0;
    probes[i] = p;
}

// perm = job indices by estimate, longest first (counting sort over 1024 classes; one workgroup of 1024 threads).
template <typename Est>
__device__ __forceinline__ void order_longest_first(Est estimate, uint32_t* __restrict__ perm, uint32_t n) {
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t max_bits;
    const uint32_t t = threadIdx.x;
    hist[t] = 0u;
    if (t == 0) max_bits = 0u;
    __syncthreads();
    uint32_t mb = 0u;
    for (uint32_t i = t; i < n; i += 1024u) { const uint32_t b = __float_as_uint(estimate(i)); mb = b > mb ? b : mb; }   // floats >= 0 order like their bits
    atomicMax(&max_bits, mb);
    __syncthreads();
    const float top = __uint_as_float(max_bits);
    const float scale = top > 0.f ? 1023.f / top : 0.f;
    auto cls = [&](uint32_t i) -> uint32_t {
        const uint32_t c = (uint32_t)(estimate(i) * scale);
        return 1023u - (c > 1023u ? 1023u : c);          // class 0 = the longest jobs
    };
    for (uint32_t i = t; i < n; i += 1024u) atomicAdd(&hist[cls(i)], 1u);
    __syncthreads();
    if (t == 0) { uint32_t acc = 0u; for (uint32_t k = 0; k < 1024u; ++k) { const uint32_t c = hist[k]; hist[k] = acc; acc += c; } }
    __syncthreads();
    for (uint32_t i = t; i < n; i += 1024u) perm[atomicAdd(&hist[cls(i)], 1u)] = i;
}

__global__ __launch_bounds__(1024) void lzf_order_by_cost_kernel(const lzf_compress_job* __restrict__ jobs,
                                                                 const lzf_job_result* __restrict__ probe_results,
                                                                 uint32_t* __restrict__ perm, uint32_t n,
                                                                 uint32_t piece, uint32_t parts) {
    order_longest_first([&](uint32_t i) -> float {
        const lzf_compress_job j = jobs[i];
        const uint64_t payload = cost_payload(j);
        const uint32_t sl = cost_sample_len(payload, piece, parts);
        if (!sl) return (float)payload * 0.1f;             // not probed: ~0.1 work units (probe batches + sequences) per byte
        uint32_t kc = 0u;                                  // probed: work units of the pieces, scaled to the payload
        for (uint32_t k = 0; k < parts; ++k) kc += probe_results[i * parts + k].reserved;
        return (float)kc * ((float)payload / (float)sl);
    }, perm, n);
}

// Decompress jobs: the work is the sequences.  Their number is estimated from three windows of the input (a token walk from an
// arbitrary byte is in step with the block's token chain within about 1 KiB: tools/seq_stats.c): 1 KiB to fall in step, then the
// tokens of the next 4 KiB counted, scaled to the input's length.  One wave per job; est[] = sequences + input_len >> len_shift (the
// bytes cost too: staging, literals).
__global__ __launch_bounds__(64) void lzf_decompress_cost_kernel(const lzf_decompress_job* __restrict__ jobs, uint32_t n, uint32_t* __restrict__ est, uint32_t len_shift) {
    constexpr uint32_t kSkip = 1024, kCount = 4096, kW = kSkip + kCount, kPad = 32;
    __shared__ __attribute__((aligned(16))) uint8_t win[3][kW + kPad];
    const uint32_t j = blockIdx.x;
    if (j >= n) return;
    const uint32_t lane = threadIdx.x & 63u;
    const lzf_decompress_job job = jobs[j];
    const uint32_t len = job.input_len > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)job.input_len;
    if (!job.input || len < 3u * (kW + kPad)) { if (lane == 0u) est[j] = len >> 4; return; }      // (short inputs: a guess; they do not decide a launch's end)
    cgu8* __restrict__ in = as_global(job.input);
    const uint32_t third = (len - (kW + kPad)) / 3u;
    for (uint32_t w = 0; w < 3u; ++w)
        for (uint32_t i = lane * 16u; i < kW + kPad; i += 1024u) {
            cgu8* g = in + w * third + i;                 // (16 bytes, any alignment)
            u32x4 v; v[0] = ld4(g); v[1] = ld4(g + 4u); v[2] = ld4(g + 8u); v[3] = ld4(g + 12u);
            *reinterpret_cast<u32x4*>(&win[w][i]) = v;
        }
    __syncthreads();
    uint32_t cnt = 0, span = 0;
    if (lane < 3u) {
        const uint8_t* b = win[lane];
        uint32_t p = 0, first = 0xFFFFFFFFu;
        while (p < kW) {
            if (p >= (lane ? kSkip : 0u)) { if (first == 0xFFFFFFFFu) first = p; ++cnt; }
            const uint32_t tok = b[p];
            uint32_t q = p + 1u, L = tok >> 4;
            if (L == 15u) { uint32_t x; do { x = q < kW + kPad ? b[q] : 0u; ++q; L += x; } while (x == 255u && q < kW + kPad); }
            q += L + 2u;
            if ((tok & 15u) == 15u) { uint32_t x; do { x = q < kW + kPad ? b[q] : 0u; ++q; } while (x == 255u && q < kW + kPad); }
            if (q <= p) break;
            p = q;
        }
        span = first == 0xFFFFFFFFu ? 0u : (p < kW + kPad ? p : kW + kPad) - first;
    }
    cnt = __builtin_amdgcn_readlane(cnt, 0) + __builtin_amdgcn_readlane(cnt, 1) + __builtin_amdgcn_readlane(cnt, 2);
    span = __builtin_amdgcn_readlane(span, 0) + __builtin_amdgcn_readlane(span, 1) + __builtin_amdgcn_readlane(span, 2);
    if (lane == 0u) est[j] = (span ? (uint32_t)(((uint64_t)len * cnt) / span) : len >> 4) + (len_shift < 32u ? len >> len_shift : 0u);
}
__global__ __launch_bounds__(1024) void lzf_order_by_estimate_kernel(const uint32_t* __restrict__ est, uint32_t* __restrict__ perm, uint32_t n) {
    order_longest_first([&](uint32_t i) -> float { return (float)est[i]; }, perm, n);
}

// Decompress jobs by compressed bytes (the order when no scratch for the estimates is to be had).
__global__ __launch_bounds__(1024) void lzf_order_by_input_len_kernel(const lzf_decompress_job* __restrict__ jobs,
                                                                      uint32_t* __restrict__ perm, uint32_t n) {
    order_longest_first([&](uint32_t i) -> float { return (float)jobs[i].input_len; }, perm, n);
}

}  // namespace lzf
