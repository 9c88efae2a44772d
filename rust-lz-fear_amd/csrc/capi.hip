// capi.hip — the extern "C" boundary declared in include/lzfear_hip.h.
// Plain HIP runtime calls + kernel launches; no CPU codec anywhere in this file: without a
// usable HIP device every entry point fails with LZF_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "kernels.h"

namespace {

thread_local std::string g_last_error;

int fail_hip(hipError_t e, const char* what) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    g_last_error = buf;
    return LZF_E_HIP;
}
#define HIP_TRY(expr)                                          \
    do {                                                       \
        hipError_t e__ = (expr);                               \
        if (e__ != hipSuccess) return fail_hip(e__, #expr);    \
    } while (0)

int ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_last_error = "no HIP device available (the lz-fear HIP codec has no CPU fallback)";
        (void)hipGetLastError();
        return LZF_E_NO_DEVICE;
    }
    return n;
}

// Kernel variant used by lzf_decompress_batch.  Tuning / A-B knob only (every variant implements the same
// contract): LZF_DECOMPRESS_KERNEL = wave (first generation, one sequence at a time) or one of the names in
// LZF_DECOMPRESS_VARIANTS (kernels.h).  Unknown names select the default.
enum { kVariantAuto = -1, kVariantWave = 0, kVariantFirstBatched = 1, kVariantFirstWindowed = 100, kVariantFirstPaired = 200, kVariantFirstWalk = 300, kVariantFirstV4 = 400, kVariantFirstV5 = 500, kVariantFirstV6 = 600 };
// Variant by name; "auto" (the default) = the producer/consumer pair kernel, with 48-byte regions while every block's
// workgroup is resident at once (lowest latency per block: the copy stage is the critical path, the parse rides along)
// and 24-byte regions beyond that (smaller LDS footprint, more blocks in flight); batches of more than eight times that
// many blocks (small blocks, typically) go to the one-wave staged16 kernel, which has no per-block pipeline to fill.
static int variant_by_name(const char* e) {
    if (!strcmp(e, "auto")) return kVariantAuto;
    if (!strcmp(e, "wave")) return kVariantWave;
    int id = kVariantFirstBatched;
#define LZF_NAME(NAME, R, S_, T, ST) if (!strcmp(e, #NAME)) return id; ++id;
    LZF_DECOMPRESS_VARIANTS(LZF_NAME)
#undef LZF_NAME
    id = kVariantFirstWindowed;
#define LZF_NAMEW(NAME, RG, R_, W_) if (!strcmp(e, #NAME)) return id; ++id;
    LZF_WINDOWED_VARIANTS(LZF_NAMEW)
#undef LZF_NAMEW
    id = kVariantFirstPaired;
#define LZF_NAMEP(NAME, RG, S_, T) if (!strcmp(e, #NAME)) return id; ++id;
    LZF_PAIRED_VARIANTS(LZF_NAMEP)
#undef LZF_NAMEP
    id = kVariantFirstWalk;
#define LZF_NAMEK(NAME, RG, S_, T) if (!strcmp(e, #NAME)) return id; ++id;
    LZF_WALK_VARIANTS(LZF_NAMEK)
#undef LZF_NAMEK
    id = kVariantFirstV4;
#define LZF_NAME4(NAME, W_, S_, T, P) if (!strcmp(e, #NAME)) return id; ++id;
    LZF_V4_VARIANTS(LZF_NAME4)
#undef LZF_NAME4
    id = kVariantFirstV5;
#define LZF_NAME5(NAME, W_, S_, ST) if (!strcmp(e, #NAME)) return id; ++id;
    LZF_V5_VARIANTS(LZF_NAME5)
#undef LZF_NAME5
    id = kVariantFirstV6;
#define LZF_NAME6(NAME, W_, S_, ST) if (!strcmp(e, #NAME)) return id; ++id;
    LZF_V6_VARIANTS(LZF_NAME6)
#undef LZF_NAME6
    return kVariantAuto;                       // unknown names select the default
}
uint32_t cu_count() {
    static const uint32_t n = [] {
        hipDeviceProp_t p; int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256u;
        return (uint32_t)p.multiProcessorCount;
    }();
    return n;
}
int decompress_variant(uint32_t n_jobs) {
    static const int v = [] { const char* e = getenv("LZF_DECOMPRESS_KERNEL"); return (e && *e) ? variant_by_name(e) : (int)kVariantAuto; }();
    if (v != kVariantAuto) return v;
    static const int small = variant_by_name("paired48"), large = variant_by_name("paired24"), huge = variant_by_name("staged16");
    static const uint32_t resident48 = [] {     // workgroups of the 48-byte form one device holds: 8 per CU (20 KB of LDS each)
        hipDeviceProp_t p; int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 2048u;
        return 8u * (uint32_t)p.multiProcessorCount;
    }();
    if (n_jobs <= resident48) return small;
    return n_jobs <= 8u * resident48 ? large : huge;   // very many (hence small) blocks: one wave per block, no pipeline fill / drain
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// RAII device buffer for the *_host helpers
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace

extern "C" {

int lzf_abi_version(void) { return LZFEAR_ABI_VERSION; }
const char* lzf_last_error(void) { return g_last_error.c_str(); }
int lzf_device_count(void) { return ensure_device(); }

int lzf_compress_batch(const lzf_compress_job* d_jobs, lzf_job_result* d_results, uint32_t n_jobs,
                       uint32_t table_kinds, void* hip_stream) {
    if (n_jobs == 0) return LZF_OK;
    if (!d_jobs || !d_results) { g_last_error = "lzf_compress_batch: NULL job/result array"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (table_kinds == 0) table_kinds = LZF_KINDS_U32 | LZF_KINDS_U16;
    // U32 jobs with a fresh or read-only template table go to the compact-table kernel (18 instead of 10 waves per CU);
    // LZF_COMPRESS_KERNEL=general keeps everything on the general kernel (A/B knob, same output).
    static const uint32_t use_compact = [] { const char* e = getenv("LZF_COMPRESS_KERNEL"); return (e && !strcmp(e, "general")) ? 0u : 1u; }();
    // More jobs than the chip holds at once (18 one-wave jobs per CU): probe their cost and launch the longest first, or the
    // launch ends with a few long jobs running alone (aux_kernels.hip).  LZF_COMPRESS_ORDER=natural keeps the caller's order, =always orders every batch.
    static const uint32_t use_order = [] { const char* e = getenv("LZF_COMPRESS_ORDER"); return !e ? 1u : !strcmp(e, "natural") ? 0u : !strcmp(e, "always") ? 2u : 1u; }();
    uint32_t* perm = nullptr;
    void* scratch = nullptr;
    if (use_order && use_compact && (table_kinds & LZF_KINDS_U32) && (use_order == 2u || n_jobs > 18u * cu_count())) {     // ("always": test knob)
        const uint32_t piece = 65536u, parts = 1u;      // one 64 KiB piece from the middle of each payload (more or smaller pieces order no better)
        const size_t n_probes = (size_t)n_jobs * parts;
        const size_t res_off = align_up(sizeof(lzf_compress_job) * n_probes, 256);
        const size_t perm_off = res_off + align_up(sizeof(lzf_job_result) * n_probes, 256);
        // (the order is an optimisation: without scratch memory the batch simply runs in the caller's order)
        if (hipMallocAsync(&scratch, perm_off + sizeof(uint32_t) * (size_t)n_jobs, st) != hipSuccess) { (void)hipGetLastError(); scratch = nullptr; }
        if (scratch) {
            lzf_compress_job* probes = reinterpret_cast<lzf_compress_job*>(scratch);
            lzf_job_result* pres = reinterpret_cast<lzf_job_result*>(static_cast<uint8_t*>(scratch) + res_off);
            perm = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(scratch) + perm_off);
            hipLaunchKernelGGL(lzf::lzf_cost_probe_jobs_kernel, dim3((uint32_t)((n_probes + 255u) / 256u)), dim3(256), 0, st, d_jobs, probes, n_jobs, piece, parts);
            hipLaunchKernelGGL(lzf::lzf_compress_compact_kernel<true>, dim3((uint32_t)n_probes), dim3(64), 0, st, probes, pres, (uint32_t)n_probes, (const uint32_t*)nullptr);
            hipLaunchKernelGGL(lzf::lzf_order_by_cost_kernel, dim3(1), dim3(1024), 0, st, d_jobs, pres, perm, n_jobs, piece, parts);
        }
    }
    if (table_kinds & LZF_KINDS_U32) {
#ifdef LZF_DBG_DRY_MAIN      // analysis: results[].reserved = probe batches + sequences of the whole job (no output)
        if (use_compact) hipLaunchKernelGGL(lzf::lzf_compress_compact_kernel<true>, dim3(n_jobs), dim3(64), 0, st, d_jobs, d_results, n_jobs, (const uint32_t*)perm);
#else
        if (use_compact) hipLaunchKernelGGL(lzf::lzf_compress_compact_kernel<false>, dim3(n_jobs), dim3(64), 0, st, d_jobs, d_results, n_jobs, (const uint32_t*)perm);
#endif
        hipLaunchKernelGGL(lzf::lzf_compress_wave_kernel<LZF_TABLE_U32>, dim3(n_jobs), dim3(64), 0, st, d_jobs, d_results, n_jobs, use_compact, (const uint32_t*)perm);
    }
    if (table_kinds & LZF_KINDS_U16)
        hipLaunchKernelGGL(lzf::lzf_compress_wave_kernel<LZF_TABLE_U16>, dim3(n_jobs), dim3(64), 0, st, d_jobs, d_results, n_jobs, 0u, (const uint32_t*)perm);
    if (scratch) HIP_TRY(hipFreeAsync(scratch, st));
    HIP_TRY(hipGetLastError());
    return LZF_OK;
}

int lzf_decompress_batch(const lzf_decompress_job* d_jobs, lzf_job_result* d_results, uint32_t n_jobs, void* hip_stream) {
    if (n_jobs == 0) return LZF_OK;
    if (!d_jobs || !d_results) { g_last_error = "lzf_decompress_batch: NULL job/result array"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int variant = decompress_variant(n_jobs);
    // More blocks than the chip holds at once: longest (most compressed bytes) first, so that the launch does not end with a
    // few long jobs running alone.  LZF_DECOMPRESS_ORDER=natural keeps the caller's order, =always orders every batch.
    static const uint32_t use_order = [] { const char* e = getenv("LZF_DECOMPRESS_ORDER"); return !e ? 1u : !strcmp(e, "natural") ? 0u : !strcmp(e, "always") ? 2u : 1u; }();
    uint32_t* perm = nullptr;
    if ((use_order == 2u || (use_order == 1u && n_jobs > 8u * cu_count())) && (variant < kVariantFirstWindowed || variant >= kVariantFirstPaired)) {   // (not the windowed analysis variants)
        if (hipMallocAsync(reinterpret_cast<void**>(&perm), sizeof(uint32_t) * (size_t)n_jobs, st) != hipSuccess) { (void)hipGetLastError(); perm = nullptr; }   // (then: the caller's order)
        if (perm) hipLaunchKernelGGL(lzf::lzf_order_by_input_len_kernel, dim3(1), dim3(1024), 0, st, d_jobs, perm, n_jobs);
    }
    const uint32_t* cperm = perm;
    if (variant == kVariantWave) {
        hipLaunchKernelGGL(lzf::lzf_decompress_wave_kernel, dim3(n_jobs), dim3(64), 0, st, d_jobs, d_results, n_jobs, cperm);
    } else if (variant < kVariantFirstWindowed) {
        int id = kVariantFirstBatched;
#define LZF_LAUNCH(NAME, R, S_, T, ST) \
        if (variant == id++) hipLaunchKernelGGL(HIP_KERNEL_NAME(lzf::lzf_decompress_batched_kernel<R, S_, T, ST>), dim3(n_jobs), dim3(64), 0, st, d_jobs, d_results, n_jobs, cperm);
        LZF_DECOMPRESS_VARIANTS(LZF_LAUNCH)
#undef LZF_LAUNCH
    } else if (variant >= kVariantFirstV6) {
        // Parse and copy as two launches per slice of the batch.  The parse of slice s + 1 runs on a side stream while the caller's
        // stream copies slice s (the parse is bound by dependent-load latency, the copy by instruction issue: they overlap well).
        // The token lists and chunk tables of a slice live in stream-ordered scratch sized from the jobs' compressed sizes, which
        // only the device knows: one small device -> host copy (and one wait on the side stream) per call.
        static const uint32_t parse_dyn_lds = [] { const char* e = getenv("LZF_V6_PARSE_WAVES"); const long v = e ? atol(e) : 0; return v > 0 ? (uint32_t)(160u * 1024u / (uint32_t)v - 8192u) : 0u; }();
        static const uint32_t kParts = [] { const char* e = getenv("LZF_V6_PARTS"); const long v = e ? atol(e) : 0; return v > 0 ? (uint32_t)v : 1u; }();
        static thread_local hipStream_t side = nullptr;
        if (!side) {
            HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            // keep the stream-ordered pool's memory between calls (the default gives it back at every synchronisation point, and
            // mapping gigabytes of scratch again costs far more than the kernels)
            int dev = 0; hipMemPool_t pool = nullptr;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
                uint64_t keep = ~0ull;
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            }
            (void)hipGetLastError();
        }
        uint32_t chunk_bytes = 0;
        { int id = kVariantFirstV6;
#define LZF_CHUNK6(NAME, W_, S_, ST) if (variant == id++) chunk_bytes = 64u * S_;
          LZF_V6_VARIANTS(LZF_CHUNK6)
#undef LZF_CHUNK6
        }
        // the parts interleave the launch order (part s = launch indices s, s + K, ...): every part gets the same mix of long and short jobs
        const uint32_t n_slices = n_jobs >= 4096u * kParts ? kParts : 1u;
        hipEvent_t ev_in = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev_in, st));                      // the jobs (and the launch order) are ready on the caller's stream
        HIP_TRY(hipStreamWaitEvent(side, ev_in, 0));
        uint64_t* offs = nullptr;                                 // per slice: cnt + 1 token offsets, cnt + 1 table offsets
        HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&offs), sizeof(uint64_t) * 2u * ((size_t)n_jobs + 2u * n_slices), side));
        std::vector<uint64_t*> tok_off(n_slices), tab_off(n_slices);
        std::vector<uint64_t> totals(2u * n_slices, 0);
        { uint64_t* p = offs;
          for (uint32_t s = 0; s < n_slices; ++s) {
              const uint32_t base = s, cnt = (n_jobs - s + n_slices - 1u) / n_slices;
              tok_off[s] = p; p += cnt + 1u; tab_off[s] = p; p += cnt + 1u;
              hipLaunchKernelGGL(lzf::lzf_v6_plan_kernel, dim3(1), dim3(1024), 0, side, d_jobs, n_jobs, base, n_slices, cnt, cperm, chunk_bytes, tok_off[s], tab_off[s]);
              HIP_TRY(hipMemcpyAsync(&totals[2u * s], tok_off[s] + cnt, sizeof(uint64_t), hipMemcpyDeviceToHost, side));
              HIP_TRY(hipMemcpyAsync(&totals[2u * s + 1u], tab_off[s] + cnt, sizeof(uint64_t), hipMemcpyDeviceToHost, side));
          } }
        HIP_TRY(hipStreamSynchronize(side));
        // (all slices' scratch up front: an allocation that re-used memory freed on the other stream would serialise the two)
        std::vector<uint32_t*> scratch(n_slices, nullptr);
        for (uint32_t s = 0; s < n_slices; ++s)
            HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&scratch[s]), sizeof(uint32_t) * (size_t)(totals[2u * s] + totals[2u * s + 1u] + 64u), side));
        for (uint32_t s = 0; s < n_slices; ++s) {
            const uint32_t base = s, cnt = (n_jobs - s + n_slices - 1u) / n_slices;
            uint32_t* toks_all = scratch[s]; uint32_t* tabs_all = scratch[s] + totals[2u * s];
            hipEvent_t ev_parsed = nullptr;
            HIP_TRY(hipEventCreateWithFlags(&ev_parsed, hipEventDisableTiming));
            int id = kVariantFirstV6;
#define LZF_LAUNCH6(NAME, W_, S_, ST) \
            if (variant == id++) { \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(lzf::lzf_v6_parse_kernel<S_, ST>), dim3(cnt), dim3(64), parse_dyn_lds, side, d_jobs, n_jobs, base, n_slices, cperm, tok_off[s], tab_off[s], toks_all, tabs_all); \
                HIP_TRY(hipEventRecord(ev_parsed, side)); \
                HIP_TRY(hipStreamWaitEvent(st, ev_parsed, 0)); \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(lzf::lzf_v6_copy_kernel<W_>), dim3(cnt), dim3(64), 0, st, d_jobs, d_results, n_jobs, base, n_slices, cperm, tok_off[s], tab_off[s], (const uint32_t*)toks_all, (const uint32_t*)tabs_all); \
            }
            LZF_V6_VARIANTS(LZF_LAUNCH6)
#undef LZF_LAUNCH6
            HIP_TRY(hipEventDestroy(ev_parsed));
        }
        for (uint32_t s = 0; s < n_slices; ++s) HIP_TRY(hipFreeAsync(scratch[s], st));
        HIP_TRY(hipFreeAsync(offs, st));
        HIP_TRY(hipEventDestroy(ev_in));
    } else if (variant >= kVariantFirstV5) {
        // the chunk token lists live in stream-ordered global scratch: 2 lists per workgroup of a launch
        uint32_t words = 0;
        { int id = kVariantFirstV5;
#define LZF_WORDS5(NAME, W_, S_, ST) if (variant == id++) words = 2u * LZF_V5_LISTWORDS(S_);
          LZF_V5_VARIANTS(LZF_WORDS5)
#undef LZF_WORDS5
        }
        const uint32_t kSlice = 16384;           // jobs per launch: bounds the scratch area
        const uint32_t slots = n_jobs < kSlice ? n_jobs : kSlice;
        uint32_t* scratch = nullptr;
        HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&scratch), (size_t)slots * words * sizeof(uint32_t), st));
        for (uint32_t base = 0; base < n_jobs; base += kSlice) {
            const uint32_t cnt = n_jobs - base < kSlice ? n_jobs - base : kSlice;
            int id = kVariantFirstV5;
#define LZF_LAUNCH5(NAME, W_, S_, ST) \
            if (variant == id++) hipLaunchKernelGGL(HIP_KERNEL_NAME(lzf::lzf_decompress_v5_kernel<W_, S_, ST>), dim3(cnt), dim3(128), 0, st, d_jobs, d_results, n_jobs, cperm, scratch, base);
            LZF_V5_VARIANTS(LZF_LAUNCH5)
#undef LZF_LAUNCH5
        }
        HIP_TRY(hipFreeAsync(scratch, st));
    } else if (variant >= kVariantFirstV4) {
        int id = kVariantFirstV4;
#define LZF_LAUNCH4(NAME, W_, S_, T, P) \
        if (variant == id++) hipLaunchKernelGGL(HIP_KERNEL_NAME(lzf::lzf_decompress_v4_kernel<W_, S_, T, P>), dim3(n_jobs), dim3(128), 0, st, d_jobs, d_results, n_jobs, cperm);
        LZF_V4_VARIANTS(LZF_LAUNCH4)
#undef LZF_LAUNCH4
    } else if (variant >= kVariantFirstWalk) {
        int id = kVariantFirstWalk;
#define LZF_LAUNCHK(NAME, RG, S_, T) \
        if (variant == id++) hipLaunchKernelGGL(HIP_KERNEL_NAME(lzf::lzf_decompress_walk_kernel<RG, S_, T>), dim3(n_jobs), dim3(128), 0, st, d_jobs, d_results, n_jobs, cperm);
        LZF_WALK_VARIANTS(LZF_LAUNCHK)
#undef LZF_LAUNCHK
    } else if (variant >= kVariantFirstPaired) {
        int id = kVariantFirstPaired;
#define LZF_LAUNCHP(NAME, RG, S_, T) \
        if (variant == id++) hipLaunchKernelGGL(HIP_KERNEL_NAME(lzf::lzf_decompress_paired_kernel<RG, S_, T>), dim3(n_jobs), dim3(128), 0, st, d_jobs, d_results, n_jobs, cperm);
        LZF_PAIRED_VARIANTS(LZF_LAUNCHP)
#undef LZF_LAUNCHP
    } else {
        // windowed kernels keep each block's token list in a stream-ordered scratch area (freed when the kernels are done)
        uint32_t stride = 0;
        { int id = kVariantFirstWindowed;
#define LZF_STRIDE(NAME, RG, R_, W_) if (variant == id++) stride = LZF_WINDOWED_STRIDE(R_);
          LZF_WINDOWED_VARIANTS(LZF_STRIDE)
#undef LZF_STRIDE
        }
        const uint32_t kSlice = 16384;           // jobs per launch: bounds the scratch area to ~0.4 GB
        const uint32_t slots = n_jobs < kSlice ? n_jobs : kSlice;
        uint16_t* scratch = nullptr;
        HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&scratch), (size_t)slots * stride * sizeof(uint16_t), st));
        for (uint32_t base = 0; base < n_jobs; base += kSlice) {
            const uint32_t cnt = n_jobs - base < kSlice ? n_jobs - base : kSlice;
            int id = kVariantFirstWindowed;
#define LZF_LAUNCHW(NAME, RG, R_, W_) \
            if (variant == id++) hipLaunchKernelGGL(HIP_KERNEL_NAME(lzf::lzf_decompress_windowed_kernel<RG, R_, W_>), dim3(cnt), dim3(64), 0, st, d_jobs + base, d_results + base, cnt, scratch, stride);
            LZF_WINDOWED_VARIANTS(LZF_LAUNCHW)
#undef LZF_LAUNCHW
        }
        HIP_TRY(hipFreeAsync(scratch, st));
    }
    if (perm) HIP_TRY(hipFreeAsync(perm, st));
    HIP_TRY(hipGetLastError());
    return LZF_OK;
}

int lzf_table_seed_from_dictionary(lzf_u32_table* d_table, const uint8_t* d_dict, uint64_t dict_len, void* hip_stream) {
    if (!d_table || (!d_dict && dict_len)) { g_last_error = "lzf_table_seed_from_dictionary: NULL argument"; return LZF_E_INVALID; }
    if (dict_len > 0xFFFFFFFFull) { g_last_error = "dictionary beyond u32 positions (reference panics, mod.rs:67)"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    HIP_TRY(hipMemsetAsync(d_table, 0, sizeof(lzf_u32_table), st));   // U32Table::default()
    if (dict_len >= 8) {
        const uint64_t count = (dict_len - 8) / 3 + 1;
        uint32_t blocks = (uint32_t)((count + 255) / 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(lzf::lzf_seed_table_kernel, dim3(blocks), dim3(256), 0, st, d_table, d_dict, dict_len);
        HIP_TRY(hipGetLastError());
    }
    return LZF_OK;
}

int lzf_table_offset(void* d_table, uint32_t table_kind, uint64_t add, void* hip_stream) {
    if (!d_table || table_kind > LZF_TABLE_U16) { g_last_error = "lzf_table_offset: bad argument"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    hipLaunchKernelGGL(lzf::lzf_table_offset_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(hip_stream), d_table, table_kind, add);
    HIP_TRY(hipGetLastError());
    return LZF_OK;
}

int lzf_table_offset_batch(void* const* d_tables, const uint64_t* d_adds, uint32_t n, uint32_t table_kind, void* hip_stream) {
    if (n == 0) return LZF_OK;
    if (!d_tables || !d_adds || table_kind > LZF_TABLE_U16) { g_last_error = "lzf_table_offset_batch: bad argument"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    hipLaunchKernelGGL(lzf::lzf_table_offset_batch_kernel, dim3((n + 255u) / 256u), dim3(256), 0, static_cast<hipStream_t>(hip_stream), d_tables, d_adds, n, table_kind);
    HIP_TRY(hipGetLastError());
    return LZF_OK;
}

int lzf_chain_decompress_step(const lzf_chain_step* d_steps, lzf_chain_state* d_state, uint32_t n_streams,
                              lzf_decompress_job* d_jobs, const lzf_job_result* d_results, void* hip_stream) {
    if (n_streams == 0) return LZF_OK;
    if (!d_steps || !d_state || !d_jobs || !d_results) { g_last_error = "lzf_chain_decompress_step: NULL argument"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    hipLaunchKernelGGL(lzf::lzf_chain_decompress_step_kernel, dim3(n_streams), dim3(256), 0, static_cast<hipStream_t>(hip_stream), d_steps, d_state, n_streams, d_jobs, d_results);
    HIP_TRY(hipGetLastError());
    return LZF_OK;
}

int lzf_copy_ranges(const uint8_t* const* d_src, uint8_t* const* d_dst, const uint64_t* d_len, uint32_t n, uint64_t max_len, void* hip_stream) {
    if (n == 0 || max_len == 0) return LZF_OK;
    if (!d_src || !d_dst || !d_len) { g_last_error = "lzf_copy_ranges: NULL argument"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    const uint64_t pieces = (max_len + 65535ull) / 65536ull;
    if (pieces > 0x7FFFFFFFull) { g_last_error = "lzf_copy_ranges: max_len too large"; return LZF_E_INVALID; }
    for (uint32_t base = 0; base < n; base += 65535u) {              // grid.y is limited to 65535
        const uint32_t cnt = n - base < 65535u ? n - base : 65535u;
        hipLaunchKernelGGL(lzf::lzf_copy_ranges_kernel, dim3((uint32_t)pieces, cnt), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                           d_src + base, d_dst + base, d_len + base, cnt);
    }
    HIP_TRY(hipGetLastError());
    return LZF_OK;
}

int lzf_xxh32_batch(const uint8_t* const* d_ptrs, const uint64_t* d_lens, uint32_t* d_out, uint32_t n, void* hip_stream) {
    if (n == 0) return LZF_OK;
    if (!d_ptrs || !d_lens || !d_out) { g_last_error = "lzf_xxh32_batch: NULL argument"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    hipLaunchKernelGGL(lzf::lzf_xxh32_kernel, dim3((n + 15) / 16), dim3(64), 0, static_cast<hipStream_t>(hip_stream), d_ptrs, d_lens, d_out, n);
    HIP_TRY(hipGetLastError());
    return LZF_OK;
}

// ---------------------------------------------------------------------------------------
// host-buffer helpers: stage -> launch -> copy back.  Synchronous.
// ---------------------------------------------------------------------------------------
int lzf_compress_batch_host(const lzf_compress_job* jobs, lzf_job_result* results, uint32_t n_jobs) {
    if (n_jobs == 0) return LZF_OK;
    if (!jobs || !results) { g_last_error = "lzf_compress_batch_host: NULL argument"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    // layout of one staging slab: [inputs | outputs | tables]
    std::vector<size_t> in_off(n_jobs), out_off(n_jobs), tab_off(n_jobs);
    size_t total = 0;
    uint32_t kinds = 0;
    for (uint32_t i = 0; i < n_jobs; ++i) {
        if (jobs[i].table_kind > LZF_TABLE_U16) { g_last_error = "bad table_kind"; return LZF_E_INVALID; }
        kinds |= jobs[i].table_kind == LZF_TABLE_U32 ? LZF_KINDS_U32 : LZF_KINDS_U16;
        in_off[i] = total; total = align_up(total + jobs[i].input_len, 256);
    }
    for (uint32_t i = 0; i < n_jobs; ++i) { out_off[i] = total; total = align_up(total + jobs[i].out_cap, 256); }
    for (uint32_t i = 0; i < n_jobs; ++i) {
        tab_off[i] = total;
        if (jobs[i].table) total = align_up(total + sizeof(lzf_u32_table), 256);   // both table structs are 16392 B
    }
    static_assert(sizeof(lzf_u32_table) == sizeof(lzf_u16_table), "table structs share a slab slot size");
    DevBuf slab, djobs, dres;
    HIP_TRY(slab.alloc(total));
    HIP_TRY(djobs.alloc(sizeof(lzf_compress_job) * n_jobs));
    HIP_TRY(dres.alloc(sizeof(lzf_job_result) * n_jobs));
    uint8_t* base = slab.as<uint8_t>();
    std::vector<lzf_compress_job> dj(jobs, jobs + n_jobs);
    for (uint32_t i = 0; i < n_jobs; ++i) {
        if (jobs[i].input_len) HIP_TRY(hipMemcpy(base + in_off[i], jobs[i].input, jobs[i].input_len, hipMemcpyHostToDevice));
        dj[i].input = base + in_off[i];
        dj[i].out = base + out_off[i];
        if (jobs[i].table) {
            HIP_TRY(hipMemcpy(base + tab_off[i], jobs[i].table, sizeof(lzf_u32_table), hipMemcpyHostToDevice));
            dj[i].table = base + tab_off[i];
        }
    }
    HIP_TRY(hipMemcpy(djobs.p, dj.data(), sizeof(lzf_compress_job) * n_jobs, hipMemcpyHostToDevice));
    rc = lzf_compress_batch(djobs.as<lzf_compress_job>(), dres.as<lzf_job_result>(), n_jobs, kinds, nullptr);
    if (rc != LZF_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(results, dres.p, sizeof(lzf_job_result) * n_jobs, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n_jobs; ++i) {
        if (results[i].status == LZF_OK && results[i].out_len)
            HIP_TRY(hipMemcpy(jobs[i].out, base + out_off[i], results[i].out_len, hipMemcpyDeviceToHost));
        if (jobs[i].table && !(jobs[i].flags & LZF_CJOB_TABLE_READONLY) && results[i].status != LZF_CONTRACT)
            HIP_TRY(hipMemcpy(jobs[i].table, base + tab_off[i], sizeof(lzf_u32_table), hipMemcpyDeviceToHost));
    }
    return LZF_OK;
}

int lzf_decompress_batch_host(const lzf_decompress_job* jobs, lzf_job_result* results, uint32_t n_jobs) {
    if (n_jobs == 0) return LZF_OK;
    if (!jobs || !results) { g_last_error = "lzf_decompress_batch_host: NULL argument"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    std::vector<size_t> in_off(n_jobs), pre_off(n_jobs), out_off(n_jobs);
    size_t total = 0;
    for (uint32_t i = 0; i < n_jobs; ++i) { in_off[i] = total; total = align_up(total + jobs[i].input_len, 256); }
    for (uint32_t i = 0; i < n_jobs; ++i) { pre_off[i] = total; total = align_up(total + jobs[i].prefix_len, 256); }
    for (uint32_t i = 0; i < n_jobs; ++i) { out_off[i] = total; total = align_up(total + jobs[i].out_cap, 256); }
    DevBuf slab, djobs, dres;
    HIP_TRY(slab.alloc(total));
    HIP_TRY(djobs.alloc(sizeof(lzf_decompress_job) * n_jobs));
    HIP_TRY(dres.alloc(sizeof(lzf_job_result) * n_jobs));
    uint8_t* base = slab.as<uint8_t>();
    std::vector<lzf_decompress_job> dj(jobs, jobs + n_jobs);
    for (uint32_t i = 0; i < n_jobs; ++i) {
        if (jobs[i].out_existing_len > jobs[i].out_cap) { g_last_error = "out_existing_len > out_cap"; return LZF_E_INVALID; }
        if (jobs[i].input_len) HIP_TRY(hipMemcpy(base + in_off[i], jobs[i].input, jobs[i].input_len, hipMemcpyHostToDevice));
        if (jobs[i].prefix_len) HIP_TRY(hipMemcpy(base + pre_off[i], jobs[i].prefix, jobs[i].prefix_len, hipMemcpyHostToDevice));
        if (jobs[i].out_existing_len) HIP_TRY(hipMemcpy(base + out_off[i], jobs[i].out, jobs[i].out_existing_len, hipMemcpyHostToDevice));
        dj[i].input = base + in_off[i];
        dj[i].prefix = base + pre_off[i];
        dj[i].out = base + out_off[i];
    }
    HIP_TRY(hipMemcpy(djobs.p, dj.data(), sizeof(lzf_decompress_job) * n_jobs, hipMemcpyHostToDevice));
    rc = lzf_decompress_batch(djobs.as<lzf_decompress_job>(), dres.as<lzf_job_result>(), n_jobs, nullptr);
    if (rc != LZF_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(results, dres.p, sizeof(lzf_job_result) * n_jobs, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n_jobs; ++i) {
        uint64_t n = results[i].out_len;
        if (n > jobs[i].out_cap) n = jobs[i].out_cap;
        if (n > jobs[i].out_existing_len)
            HIP_TRY(hipMemcpy(jobs[i].out + jobs[i].out_existing_len, base + out_off[i] + jobs[i].out_existing_len,
                              n - jobs[i].out_existing_len, hipMemcpyDeviceToHost));
    }
    return LZF_OK;
}

}  // extern "C"
