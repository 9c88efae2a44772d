// capi.hip — the extern "C" boundary declared in include/lzfear_hip.h.
// Plain HIP runtime calls + kernel launches; no CPU codec anywhere in this file: without a
// usable HIP device every entry point fails with LZF_E_NO_DEVICE.  The product library reads no
// environment variables; the A/B knobs of the analysis build live in analysis/capi_analysis.inc.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>
#include "kernels.h"
#include "host_staging.h"

namespace {

thread_local std::string g_last_error;
thread_local const char* g_last_decompress = "";      // what the last lzf_decompress_batch of this thread launched (lzf_last_decompress_launch)
thread_local const char* g_last_compress = "";        // ... and the last lzf_compress_batch

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
// a kernel launch and its launch status (a failed launch must not pass for an empty result array)
#define LAUNCH(...)                                            \
    do {                                                       \
        hipLaunchKernelGGL(__VA_ARGS__);                       \
        HIP_TRY(hipGetLastError());                            \
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

// The geometry every dispatch threshold below is derived from: compute units and LDS bytes per CU of the current device, as the
// runtime reports them (MI355X: 256 and 160 KiB).  The thresholds are functions of those two numbers (the segmented pipeline's
// batch limit additionally of its rank kernels' 1024-job workgroup, kSegRankMax).
struct Geometry { uint32_t cu, lds, wall_khz; };      // wall_khz: rate of wall_clock64() (s_memrealtime), 100 MHz on MI355X
const Geometry& geometry() {
    static const Geometry g = [] {
        Geometry r{256u, 160u * 1024u, 100000u};
        hipDeviceProp_t p; int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) {
            r.cu = (uint32_t)p.multiProcessorCount;
            if (p.maxSharedMemoryPerMultiProcessor >= 64u * 1024u) r.lds = (uint32_t)p.maxSharedMemoryPerMultiProcessor;
            int khz = 0;
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz >= 1000) r.wall_khz = (uint32_t)khz; else (void)hipGetLastError();
        } else (void)hipGetLastError();      // (only an error of these calls is cleared, never one the caller left pending)
#ifdef LZF_ANALYSIS      // LZF_FAKE_CU=n: dispatch as if the device had n compute units (test of the derived thresholds on one device)
        if (const char* e = getenv("LZF_FAKE_CU")) { const long v = atol(e); if (v >= 1 && v <= 4096) r.cu = (uint32_t)v; }
#endif
        return r;
    }();
    return g;
}
uint32_t cu_count() { return geometry().cu; }
// workgroups of `lds_bytes` of LDS each that one CU holds
uint32_t per_cu(uint32_t lds_bytes) { const uint32_t n = geometry().lds / lds_bytes; return n ? n : 1u; }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// stream-ordered scratch of a batch call: handed back to the pool on every way out of the call (a failed launch included)
struct AsyncScratch {
    void* p = nullptr; hipStream_t st = nullptr;
    ~AsyncScratch() { if (p) { (void)hipFreeAsync(p, st); } }
    hipError_t release() { void* q = p; p = nullptr; return q ? hipFreeAsync(q, st) : hipSuccess; }
};

// The batch calls take their scratch (launch orders, cost probes) from the stream-ordered pool.  By default the pool hands its
// memory back at every synchronisation point, so the next call allocates from the device again — which waits for whatever is
// running.  Keep what the pool has: a later hipMallocAsync on the same stream re-uses it without touching the device.
void keep_pool_memory() {
    static thread_local int done_for = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev == done_for) return;
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
        uint64_t keep = ~0ull;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    (void)hipGetLastError();
    done_for = dev;
}

// the three kernels of the product dispatch (lzf_decompress_batch)
constexpr auto k_paired48 = lzf::lzf_decompress_paired_kernel<4096, 48, 640>;
constexpr auto k_paired24 = lzf::lzf_decompress_paired_kernel<4096, 24, 384>;
constexpr auto k_staged16 = lzf::lzf_decompress_batched_kernel<4096, 16, 256, true>;
constexpr auto k_compact = lzf::lzf_compress_compact_kernel<false>;
constexpr auto k_compact_dry = lzf::lzf_compress_compact_kernel<true>;
constexpr auto k_general_u32 = lzf::lzf_compress_wave_kernel<LZF_TABLE_U32>;
constexpr auto k_general_u16 = lzf::lzf_compress_wave_kernel<LZF_TABLE_U16>;
constexpr uint32_t kTeamLds = 163840u;      // LDS of one workgroup of lzf_compress_team_kernel (lz4_compress_team.inc: team::kLdsWords * 4)
constexpr uint32_t kTeamRounds = 1u;        // batches of up to this many jobs per CU take the team kernel

// ---- the segmented pipeline (lz4_decompress_seg.hip): geometry, scratch, launches ------------------------------------
constexpr uint32_t kSegMaxIn = 4u * 1024u * 1024u + 32u * 1024u;     // a 4 MiB block at LZ4's worst case, rounded up
constexpr uint32_t kSegMinIn = 64u * 1024u;                          // smaller blocks are done sooner by one workgroup
// LDS of one workgroup of the resolve stage: its ring + flags, tickets and slack (lz4_decompress_seg.hip).  The pipeline takes
// batches of up to one block per 32 KiB ring the chip's LDS holds (MI355X: 4 per CU = 1 024 blocks; at 980 blocks 18.2 ms against 23.5
// for the pair kernel) and gives a block the largest ring that still leaves every block of the batch resident at once.
constexpr uint32_t kSegRingSlack = 8u * 1024u;
inline uint32_t seg_blocks_per_cu(uint32_t ring) { return per_cu(ring + kSegRingSlack); }
// (capped by lzf_seg_by_len_kernel / lzf_seg_order_kernel: one 1024-thread workgroup ranks the batch in a `cost[1024]` LDS array, lz4_decompress_seg.hip)
constexpr uint32_t kSegRankMax = 1024u;
inline uint32_t seg_max_jobs() { const uint32_t n = seg_blocks_per_cu(32768u) * cu_count(); return n < kSegRankMax ? n : kSegRankMax; }
inline uint32_t seg_ring_for(uint32_t n) {
    return n <= seg_blocks_per_cu(131072u) * cu_count() && geometry().lds >= 131072u + kSegRingSlack ? 131072u
         : n <= seg_blocks_per_cu(65536u) * cu_count() && geometry().lds >= 65536u + kSegRingSlack ? 65536u : 32768u;
}
constexpr uint64_t kSegRecsPerJob = 448u * 1024u;                    // arena: records per job on average (16 bytes each)

struct SegScratch {
    void* base = nullptr;
    lzf::seg_ctx ctx{};
    size_t bytes = 0;
    uint32_t* est = nullptr;       // [n] grouped calls: the jobs by sequences, most first (lzf_seg_rank_kernel)
};
inline uint32_t seg_nch_host(uint32_t len) { return len <= lzf::kSegChunk ? 1u : 1u + (len - lzf::kSegChunk + lzf::kSegStride - 1u) / lzf::kSegStride; }

// lays the scratch areas of a call out in one stream-ordered allocation; false (and nothing allocated) when the pool has no room
bool seg_alloc(SegScratch& s, const lzf_decompress_job* d_jobs, lzf_job_result* d_results, uint32_t n, uint32_t min_in, hipStream_t st, uint64_t max_in_hint = ~0ull) {
    lzf::seg_ctx& c = s.ctx;
    c.jobs = d_jobs; c.results = d_results; c.n_jobs = n;
#ifdef LZF_ANALYSIS      // LZF_SEG_FORCE=noscratch | stager | resolver: the pipeline's fall-backs, forced (tests/test_gpu_parity.py)
    { static const uint32_t force = [] { const char* e = getenv("LZF_SEG_FORCE"); return !e ? 0u : !strcmp(e, "stager") ? 1u : !strcmp(e, "resolver") ? 2u : !strcmp(e, "noscratch") ? 3u : !strcmp(e, "swait") ? 8u : 0u; }();
      if (force == 3u) return false;
      c.dbg_force = force; }
#endif
    // (a caller that knows an upper bound of its jobs' input sizes gets scratch sized for it: the job array is in HBM, the host cannot look)
    const uint32_t max_in = max_in_hint < kSegMaxIn ? (uint32_t)(max_in_hint < lzf::kSegChunk ? lzf::kSegChunk : max_in_hint) : kSegMaxIn;
    c.max_in = max_in; c.min_in = min_in;
    c.maxch = seg_nch_host(max_in);
    c.maxtile = (max_in + lzf::kSegTile - 1u) / lzf::kSegTile;
    { const uint64_t per_job = (uint64_t)max_in / 3u + 192u; c.rec_cap = (uint64_t)n * (per_job < kSegRecsPerJob ? per_job : kSegRecsPerJob); }
    // the ring of a block: 128 KiB holds every distance LZ4 can express (no read-backs from HBM) while a CU has one block,
    // 64 / 32 KiB with read-backs for the oldest few per cent of the sources beyond that
    c.ring_bytes = seg_ring_for(n);
#ifdef LZF_ANALYSIS      // LZF_SEG_RING=32768|65536|131072: a ring size whatever the batch (one block per CU with the small rings: the stager's share)
    { static const uint32_t ring = [] { const char* e = getenv("LZF_SEG_RING"); return e ? (uint32_t)atol(e) : 0u; }();
      if (ring == 32768u || ring == 65536u || ring == 131072u) c.ring_bytes = ring; }
#endif
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_st = take(sizeof(lzf::seg_job) * (size_t)n);
    const size_t o_top = take(sizeof(unsigned long long));
    const size_t o_x = take(sizeof(uint32_t) * (size_t)n * c.maxch);
    const size_t o_vf = take(sizeof(uint32_t) * (size_t)n * c.maxch);
    const size_t o_tt = take(sizeof(uint32_t) * (size_t)n * c.maxtile);
    const size_t o_to = take(sizeof(uint32_t) * (size_t)n * c.maxtile);
    const size_t o_bits = take(sizeof(uint32_t) * (size_t)n * c.maxch * lzf::kSegChunkWords);
    const size_t o_recs = take(sizeof(lzf::u32x4) * (size_t)c.rec_cap);
    const size_t o_ord = take(sizeof(uint32_t) * (size_t)n);
    const size_t o_len = take(sizeof(uint32_t) * (size_t)n);
    const size_t o_est = take(sizeof(uint32_t) * (size_t)n);
    if (hipMallocAsync(&s.base, off, st) != hipSuccess) { (void)hipGetLastError(); s.base = nullptr; return false; }
    s.bytes = off;
    uint8_t* b = static_cast<uint8_t*>(s.base);
    c.st = reinterpret_cast<lzf::seg_job*>(b + o_st);
    c.rec_top = reinterpret_cast<unsigned long long*>(b + o_top);
    c.xexit = reinterpret_cast<uint32_t*>(b + o_x);
    c.vfrom = reinterpret_cast<uint32_t*>(b + o_vf);
    c.tile_tok = reinterpret_cast<uint32_t*>(b + o_tt);
    c.tile_out = reinterpret_cast<uint32_t*>(b + o_to);
    c.bits = reinterpret_cast<uint32_t*>(b + o_bits);
    c.recs = reinterpret_cast<lzf::u32x4*>(b + o_recs);
    c.n_cu = cu_count();
    c.order = (n > c.n_cu && n <= seg_max_jobs()) ? reinterpret_cast<uint32_t*>(b + o_ord) : nullptr;      // (one block per CU: nothing to balance)
    c.by_len = (n >= (c.n_cu + 7u) / 8u && n <= seg_max_jobs()) ? reinterpret_cast<uint32_t*>(b + o_len) : nullptr;       // (MI355X: 32 jobs and more)
    c.rec_by_len = n >= (c.n_cu + 3u) / 4u ? 1u : 0u;                                                                      // (64 and more)
    c.g_off = 0u; c.g_n = n; c.grouped = 0u; c.res_prio = 0u; c.fed = 0u;
    s.est = reinterpret_cast<uint32_t*>(b + o_est);
    return true;
}
inline uint32_t seg_grid(uint32_t target, uint32_t n, uint32_t cap) {
    uint32_t g = target / n; if (g < 1u) g = 1u; if (g > cap) g = cap; return g;
}
// stages: 1 plan, 2 parse, 3 seam, 4 tilesum, 5 scan, 6 records (+ levels), 8 resolve (all when upto >= 8)
// seg_launch_prep: the stages that look at every job of the call; seg_launch_front: the chunk / tile stages of the group the context
// names (ranks g_off .. g_off + g_n); seg_launch_resolve: its resolve stage.
int seg_launch_prep(const lzf::seg_ctx& c, hipStream_t st) {
    if (c.by_len && !c.grouped) LAUNCH(lzf::lzf_seg_by_len_kernel, dim3(1), dim3(1024), 0, st, c);
    LAUNCH(lzf::lzf_seg_plan_kernel, dim3((c.n_jobs + 255u) / 256u), dim3(256), 0, st, c);
    return LZF_OK;
}
int seg_launch_front(const lzf::seg_ctx& c, uint32_t upto, hipStream_t st) {
    const uint32_t n = c.g_n;
    // workgroups per launch of the chunk / tile kernels: about one chunk and a handful of tiles each — with 8 192 / 32 768 (each
    // workgroup looping over a dozen chunks) the launches ended on their slowest loops: parse 4.3 -> 3.3 ms at 980 blocks, 1.02 -> 0.83 at 196
    uint32_t tg_parse = 1024u * c.n_cu, tg_tile = 2048u * c.n_cu;             // (MI355X: 262 144 and 524 288)
#ifdef LZF_ANALYSIS      // LZF_SEG_GRID="parse,tiles": workgroups per launch of the chunk / tile kernels (A/B of the grid sizes)
    { static const char* e = getenv("LZF_SEG_GRID"); if (e) { unsigned a = 0, b = 0; if (sscanf(e, "%u,%u", &a, &b) == 2 && a && b) { tg_parse = a; tg_tile = b; } } }
#endif
    if (upto >= 2) LAUNCH(lzf::lzf_seg_parse_kernel, dim3(seg_grid(tg_parse, n, c.maxch), n), dim3(64), 0, st, c);
    if (upto >= 3) LAUNCH(lzf::lzf_seg_seam_kernel, dim3(n), dim3(64), 0, st, c);
    if (upto >= 4) LAUNCH(lzf::lzf_seg_tilesum_kernel, dim3(seg_grid(tg_tile, n, c.maxtile), n), dim3(64), 0, st, c);
    if (upto >= 5) LAUNCH(lzf::lzf_seg_scan_kernel, dim3(n), dim3(64), 0, st, c);
    if (upto >= 6 && c.order) LAUNCH(lzf::lzf_seg_order_kernel, dim3(1), dim3(1024), 0, st, c);
    if (upto >= 6) LAUNCH(lzf::lzf_seg_records_kernel, dim3(seg_grid(tg_tile, n, c.maxtile), n), dim3(64), 0, st, c);
    return LZF_OK;
}
// the records stage of the group the context names, on its own
int seg_launch_records(const lzf::seg_ctx& c, hipStream_t st) {
    const uint32_t n = c.g_n;
    uint32_t tg_tile = 2048u * c.n_cu;
#ifdef LZF_ANALYSIS
    { static const char* e = getenv("LZF_SEG_GRID"); if (e) { unsigned a = 0, b = 0; if (sscanf(e, "%u,%u", &a, &b) == 2 && a && b) tg_tile = b; } }
#endif
    uint32_t pad = 0;
#ifdef LZF_ANALYSIS      // LZF_SEG_REC_PAD = bytes of unused LDS per workgroup of a grouped call's records stage (fewer of them resident under the resolve stages: A/B)
    { static const uint32_t e = [] { const char* v = getenv("LZF_SEG_REC_PAD"); return v ? (uint32_t)atol(v) : 0u; }(); pad = e; }
#endif
    LAUNCH(lzf::lzf_seg_records_kernel, dim3(seg_grid(tg_tile, n, c.maxtile), n), dim3(64), pad, st, c);
    return LZF_OK;
}
int seg_launch_resolve(const lzf::seg_ctx& c, hipStream_t st) {
    const uint32_t n = c.g_n;
    if (c.ring_bytes == 131072u) LAUNCH(lzf::lzf_seg_resolve_pair_kernel<131072>, dim3(n), dim3(128), 0, st, c);
    else if (c.ring_bytes == 65536u) LAUNCH(lzf::lzf_seg_resolve_pair_kernel<65536>, dim3(n), dim3(128), 0, st, c);
    else LAUNCH(lzf::lzf_seg_resolve_pair_kernel<32768>, dim3(n), dim3(128), 0, st, c);
    return LZF_OK;
}
int seg_launch(const lzf::seg_ctx& c, uint32_t upto, hipStream_t st) {
    int rc = seg_launch_prep(c, st);
    if (rc == LZF_OK) rc = seg_launch_front(c, upto, st);
    if (rc == LZF_OK && upto >= 8) rc = seg_launch_resolve(c, st);
    return rc;
}

// ---- groups: the records stage of a group runs under the resolve stage of the groups before it -------------------------------
// The resolve stage is a chain per block (one pair of wavefronts, 7 ms for a 4 MiB text block whatever the batch) and leaves most
// of the chip idle; the stages before it are throughput kernels.  A call of several hundred blocks therefore takes its last two
// stages in groups, by sequences (known once the tiles are counted: lzf_seg_rank_kernel), most first: the caller's stream
// carries every group's records stage back to back, each group's resolve stage starts on a stream of its own as soon as its
// records are written (an event), and the caller's stream waits for all of them before the pair kernel looks for jobs the
// pipeline left.  The call then takes records(first group) + resolve(longest block) instead of records(all) + resolve(longest
// block), as long as the later groups — the blocks with fewer sequences — are through their shorter resolve stages by then.
constexpr uint32_t kSegMaxGroups = 4;      // (the caller's stream + three of the library's own, made at the highest stream priority: the runtime keeps a pool of hardware queues per
                                           //  priority, so these do not share a queue with the application's streams; a fifth group measured slower: 8.9 -> 13.2 ms at 196 blocks)
struct SegGroups { uint32_t n = 1; uint32_t size[kSegMaxGroups] = {}; };
SegGroups seg_groups(uint32_t n_jobs) {
    SegGroups g; g.size[0] = n_jobs;
    // (beyond what the rank kernels' one workgroup takes — only the analysis library's forced pipeline gets here — one group, in the
    //  caller's order: ADVICE r5, the ranks of 1 024 jobs and more were never written)
    if (n_jobs > seg_max_jobs()) return g;
    const uint32_t ncu = cu_count();
    uint32_t pct[kSegMaxGroups] = {100}; uint32_t k = 1;
    if (n_jobs >= (ncu + 7u) / 8u) { pct[0] = pct[1] = pct[2] = pct[3] = 25; k = 4; }      // (MI355X: 32 jobs and more; measured 49 .. 980 blocks: quarters beat halves and thirds)
#ifdef LZF_ANALYSIS      // LZF_SEG_GROUPS="a,b,c,...": per cent of the jobs per group (A/B of the grouping; "100" = one group)
    { static const char* e = getenv("LZF_SEG_GROUPS");
      if (e) { uint32_t v[kSegMaxGroups] = {}, got = 0, sum = 0; const char* q = e;
               while (got < kSegMaxGroups && *q) { char* end = nullptr; const unsigned long x = strtoul(q, &end, 10); if (end == q) break; v[got++] = (uint32_t)x; sum += (uint32_t)x; q = *end == ',' ? end + 1 : end; if (*end != ',') break; }
               if (got >= 1 && sum == 100u && v[0]) { for (uint32_t i = 0; i < kSegMaxGroups; ++i) pct[i] = v[i]; k = got; } } }
#endif
    if (k <= 1u || n_jobs < 2u * k) return g;
    uint32_t left = n_jobs; g.n = 0;
    for (uint32_t i = 0; i < k && left; ++i) {
        uint32_t sz = i + 1u == k ? left : (uint32_t)((uint64_t)n_jobs * pct[i] / 100u);
        if (sz > left) sz = left;
        if (!sz) continue;
        g.size[g.n++] = sz; left -= sz;
    }
    if (left && g.n) g.size[g.n - 1u] += left;
    return g;
}
// the streams and events of grouped calls: made once, kept; calls enqueue under the mutex (the work itself overlaps freely)
struct SegLanes {
    std::mutex mu;
    hipStream_t s[kSegMaxGroups - 1u] = {};
    hipEvent_t front[kSegMaxGroups - 1u] = {}, done[kSegMaxGroups - 1u] = {};
    bool ok = false;
};
SegLanes* seg_lanes() {
    // (per device: a stream belongs to the device that was current when it was made — one process per GPU is the rule, but a process
    //  that moves between devices must not get another device's streams)
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static SegLanes* by_dev[kMaxDev] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(mu);
    if (!by_dev[dev]) {
        SegLanes* L = new SegLanes();
        // High-priority streams: the runtime keeps a pool of hardware queues PER PRIORITY (four each by default), so these three do not
        // end up sharing a queue with the application's own streams — a resolve stage queued behind the caller's next records stage
        // would serialise the call (measured: 49 blocks 13.4 ms instead of 8.0 with one application side stream alive) — and the
        // resolve stage, the chain of the call, is dispatched ahead of the throughput kernels.
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
        bool ok = true;
        for (uint32_t i = 0; i < kSegMaxGroups - 1u && ok; ++i)
            ok = hipStreamCreateWithPriority(&L->s[i], hipStreamNonBlocking, greatest) == hipSuccess &&
                 hipEventCreateWithFlags(&L->front[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&L->done[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        L->ok = ok;
        by_dev[dev] = L;
    }
    return by_dev[dev]->ok ? by_dev[dev] : nullptr;
}
int seg_enqueue_groups(SegScratch& s, const SegGroups& g, SegLanes& L, hipStream_t st) {
    lzf::seg_ctx c = s.ctx;
    std::lock_guard<std::mutex> lk(L.mu);
    int rc = seg_launch_prep(c, st);
    if (rc == LZF_OK) rc = seg_launch_front(c, 5u, st);         // plan .. scan over the whole call
    if (rc != LZF_OK) return rc;
    static_assert(kSegMaxGroups == 4, "lzf_seg_rank_kernel takes the group sizes as a uint4");
    LAUNCH(lzf::lzf_seg_rank_kernel, dim3(1), dim3(1024), 0, st, c, s.est, make_uint4(g.size[0], g.n > 1u ? g.size[1] : 0u, g.n > 2u ? g.size[2] : 0u, g.n > 3u ? g.size[3] : 0u));      // s.est: the jobs by sequences, most first; + every group's workgroup order
    c.by_len = s.est; c.grouped = 1u; c.res_prio = 1u;
#ifdef LZF_ANALYSIS      // LZF_SEG_PRIO=0: the resolve stage at the default issue priority (A/B)
    { static const int pr = [] { const char* e = getenv("LZF_SEG_PRIO"); return e ? atoi(e) : 1; }(); c.res_prio = pr ? 1u : 0u; }
#endif
    uint32_t off = 0, forked = 0;
    const uint32_t ticks_per_us = geometry().wall_khz / 1000u;   // (the device's wall clock: 100 per microsecond on MI355X)
    uint32_t pause_ticks = 10u * ticks_per_us;                   // 10 us (5 .. 80 us measured alike)
#ifdef LZF_ANALYSIS      // LZF_SEG_PAUSE_US: the pause between a group's records stage and the next (A/B; 0 = none)
    { static const int us = [] { const char* e = getenv("LZF_SEG_PAUSE_US"); return e ? atoi(e) : -1; }(); if (us >= 0) pause_ticks = (uint32_t)us * ticks_per_us; }
#endif
    for (uint32_t k = 0; k < g.n && rc == LZF_OK; ++k) {
        c.g_off = off; c.g_n = g.size[k]; off += g.size[k];
        rc = seg_launch_records(c, st);
        if (rc != LZF_OK) break;
        if (k + 1u < g.n) {
            if (hipEventRecord(L.front[k], st) != hipSuccess || hipStreamWaitEvent(L.s[k], L.front[k], 0) != hipSuccess) { (void)hipGetLastError(); rc = LZF_E_HIP; break; }
            rc = seg_launch_resolve(c, L.s[k]);
            ++forked;                                            // (whatever was enqueued on the lane is joined below)
            if (pause_ticks) LAUNCH(lzf::lzf_seg_pause_kernel, dim3(1), dim3(64), 0, st, pause_ticks);      // the resolve stage's workgroups first, then the next records stage
            if (hipEventRecord(L.done[k], L.s[k]) != hipSuccess) { (void)hipGetLastError(); --forked; if (rc == LZF_OK) rc = LZF_E_HIP; if (hipStreamSynchronize(L.s[k]) != hipSuccess) (void)hipGetLastError(); }
        } else {
            rc = seg_launch_resolve(c, st);                      // the last group: on the caller's stream
        }
    }
    for (uint32_t k = 0; k < forked; ++k)
        if (hipStreamWaitEvent(st, L.done[k], 0) != hipSuccess) { (void)hipGetLastError(); if (hipStreamSynchronize(L.s[k]) != hipSuccess) (void)hipGetLastError(); if (rc == LZF_OK) rc = LZF_E_HIP; }
    return rc;
}
int seg_launch_grouped(SegScratch& s, const SegGroups& g, SegLanes& L, hipStream_t st) {
    const int rc = seg_enqueue_groups(s, g, L, st);
    if (rc != LZF_OK)                                            // (a launch failed part-way: nothing may still be running on a lane when the caller frees the scratch)
        for (uint32_t k = 0; k < kSegMaxGroups - 1u; ++k) if (hipStreamSynchronize(L.s[k]) != hipSuccess) (void)hipGetLastError();
    return rc;
}
// The whole call: pipeline, then the pair kernel over what the pipeline did not finish.
int seg_decompress(const lzf_decompress_job* d_jobs, lzf_job_result* d_results, uint32_t n, uint32_t min_in, hipStream_t st, bool* used, uint64_t max_in_hint, uint32_t* ring_used) {
    SegScratch s;
    *used = false; *ring_used = 0u;
    if (max_in_hint < min_in) return LZF_OK;                                      // (no job can be in the pipeline's window)
    if (!seg_alloc(s, d_jobs, d_results, n, min_in, st, max_in_hint)) return LZF_OK;      // (no scratch: the caller launches the pair kernel over everything)
    *used = true; *ring_used = s.ctx.ring_bytes;
    const SegGroups g = seg_groups(n);
    SegLanes* lanes = g.n > 1u ? seg_lanes() : nullptr;
    int rc = lanes ? seg_launch_grouped(s, g, *lanes, st) : seg_launch(s.ctx, 8u, st);
    if (rc == LZF_OK) {
        hipLaunchKernelGGL(k_paired48, dim3(n), dim3(128), 0, st, d_jobs, d_results, n, (const uint32_t*)nullptr, (const lzf::seg_job*)s.ctx.st);
        if (hipGetLastError() != hipSuccess) rc = LZF_E_HIP;
    }
    if (hipFreeAsync(s.base, st) != hipSuccess && rc == LZF_OK) rc = LZF_E_HIP;
    if (rc != LZF_OK) g_last_error = "segmented decompress: launch failed";
    return rc;
}

// ---- the bitmap-fed kernel (lz4_decompress_fed.hip): batches beyond the segmented pipeline's ----------------------------------
// plan + parse + seam of the segmented pipeline over the whole batch (the token bit map of every block: one bit per compressed
// byte), then one wavefront per block that lists its tokens from the map and copies — the in-kernel parse of the pair kernel is
// 13.0 of its 27.8 wave-instructions per sequence, the hop parse 3.8 — then the pair kernel over whatever that left (errors,
// sizes outside the map's window, a chain that did not verify).
constexpr auto k_fed32 = lzf::lzf_decompress_fed_kernel<4096, 32, 352>;
constexpr uint32_t kFedMinInDefault = 65536u;                        // per job: smaller inputs are left to the pair kernel behind the fed kernel
// per call: a caller that bounds its inputs (lzf_decompress_batch_sized, the frame layer) keeps batches of small blocks off this path
// altogether — per job the feed costs a census of pieces, a chunk's parse and a ring re-fill: 16 384 jobs of ~32 KiB ran at 380 GiB/s
// through it and at 484 through the pair kernel (bench config5, u16_raw); 18 000 blocks of 256 KiB (inputs ~128 KiB) 12.97 against
// 12.43 ms; 4 536 blocks of 1 MiB 12.72 against 14.05.  (The bound is the call's, not the job's: with the small inputs of a batch of
// large blocks left to a second kernel behind the first, the 1 MiB call took 18.2 ms.)
constexpr uint64_t kFedMinHint = 262144u;
inline uint32_t fed_min_in() {
#ifdef LZF_ANALYSIS      // LZF_FED_MIN_IN: the smallest input the bitmap-fed kernel takes (the variant parity test opens it to every input)
    static const long v = [] { const char* e = getenv("LZF_FED_MIN_IN"); return e ? atol(e) : -1L; }();
    if (v >= 0) return (uint32_t)v;
#endif
    return kFedMinInDefault;
}
inline uint64_t fed_min_hint() {
#ifdef LZF_ANALYSIS      // (LZF_FED_MIN_IN opens the path to every call as well: the variant parity test)
    static const bool open_ = getenv("LZF_FED_MIN_IN") != nullptr;
    if (open_) return 0u;
#endif
    return kFedMinHint;
}
constexpr uint64_t kFedMaxScratch = 24ull << 30;                     // bit maps of a call: 1 bit per compressed byte of the largest job x jobs (16 / 14 with the chunks' overlap)
constexpr uint32_t kFedMaxJobs = 65535u;                             // (the chunk stage's grid has one row per job)
// Workgroups of the kernel the current device holds at once, COUNTED (lz4_decompress_fed.hip, census mode): the occupancy query
// does not know the LDS allocation granule (6 912 bytes take 7 680: 21 per CU, the query says 23), and a schedule with more slots
// than residents runs its surplus slots after the others.  Once per device and process: one launch of ~0.1 ms and a 4-byte copy
// (the one place a batch call waits for the device).
struct FedGeometry { uint32_t slots, xcc_mask; };
FedGeometry fed_geometry(hipStream_t st) {
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static FedGeometry by_dev[kMaxDev] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) { (void)hipGetLastError(); return FedGeometry{8u * cu_count(), 0u}; }
    std::lock_guard<std::mutex> lk(mu);
    if (by_dev[dev].slots) return by_dev[dev];
    FedGeometry answer{0u, 0u};
    uint32_t* d = nullptr;
    if (hipMalloc(&d, 4u * sizeof(uint32_t)) == hipSuccess) {
        const uint32_t init[4] = {0u, 0xFFFFFFFFu, 0u, 0u};
        lzf::fed_args a{}; a.census = d;
        if (hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, st) == hipSuccess) {
            hipLaunchKernelGGL(k_fed32, dim3(40u * cu_count()), dim3(64), 0, st, a);
            uint32_t got[4] = {0, 0, 0, 0};
            if (hipGetLastError() == hipSuccess && hipMemcpyAsync(got, d, sizeof got, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
                got[1] != 0xFFFFFFFFu && got[1] >= cu_count()) { answer.slots = got[1]; answer.xcc_mask = got[2]; }
        }
        (void)hipFree(d);
    }
    (void)hipGetLastError();
    if (!answer.slots) answer.slots = per_cu(7680u) * cu_count();    // (the census failed: the LDS granule's answer for 6 912 bytes; no XCD mask: jobs stay whole)
#ifdef LZF_ANALYSIS
    if (getenv("LZF_FED_VERBOSE")) fprintf(stderr, "[lzf] bitmap-fed kernel: %u workgroups resident at once (%u compute units), XCD mask 0x%x\n", answer.slots, cu_count(), answer.xcc_mask);
#endif
    by_dev[dev] = answer;
    return answer;
}
// A call (or one half of it) in two steps: FRONT = scratch + plan, parse, seam (the bit maps); BACK = the copy stage fed from them and
// the pair kernel over what is left, then the scratch goes back.  The two steps may run on different streams (the caller orders them).
struct FedCall {
    SegScratch s; const lzf_decompress_job* jobs = nullptr; lzf_job_result* res = nullptr; uint32_t n = 0; const uint32_t* perm = nullptr;
    size_t o_tick = 0, o_fst = 0; bool live = false;
};
// false: the pair kernel takes these jobs (no room for the bit maps)
int fed_front(FedCall& f, const lzf_decompress_job* d_jobs, lzf_job_result* d_results, uint32_t n, const uint32_t* perm, hipStream_t st, uint64_t max_in_hint) {
    f.jobs = d_jobs; f.res = d_results; f.n = n; f.perm = perm; f.live = false;
    lzf::seg_ctx& c = f.s.ctx;
    c = lzf::seg_ctx{};
    c.jobs = d_jobs; c.results = d_results; c.n_jobs = n;
    const uint32_t max_in = max_in_hint < kSegMaxIn ? (uint32_t)(max_in_hint < lzf::kSegChunk ? lzf::kSegChunk : max_in_hint) : kSegMaxIn;
    c.max_in = max_in; c.min_in = fed_min_in();
    c.maxch = seg_nch_host(max_in);
    c.maxtile = (max_in + lzf::kSegTile - 1u) / lzf::kSegTile;
    c.n_cu = cu_count();
    c.g_off = 0u; c.g_n = n; c.fed = 1u;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_st = take(sizeof(lzf::seg_job) * (size_t)n);
    const size_t o_top = take(sizeof(unsigned long long));
    const size_t o_x = take(sizeof(uint32_t) * (size_t)n * c.maxch);
    const size_t o_vf = take(sizeof(uint32_t) * (size_t)n * c.maxch);
    const size_t o_bits = take(sizeof(uint32_t) * (size_t)n * c.maxch * lzf::kSegChunkWords);
    f.o_tick = take(sizeof(uint32_t) * 32u * lzf::kFedTicketStride);
    f.o_fst = take(sizeof(lzf::fed_state) * (size_t)n);
    if (off > kFedMaxScratch) return LZF_OK;                         // (the pair kernel takes the call)
    if (hipMallocAsync(&f.s.base, off, st) != hipSuccess) { (void)hipGetLastError(); f.s.base = nullptr; return LZF_OK; }
    uint8_t* b = static_cast<uint8_t*>(f.s.base);
    c.st = reinterpret_cast<lzf::seg_job*>(b + o_st);
    c.rec_top = reinterpret_cast<unsigned long long*>(b + o_top);
    c.xexit = reinterpret_cast<uint32_t*>(b + o_x);
    c.vfrom = reinterpret_cast<uint32_t*>(b + o_vf);
    c.bits = reinterpret_cast<uint32_t*>(b + o_bits);
    f.live = true;
    const int rc = seg_launch(c, 3u, st);                            // plan, parse, seam
    if (rc != LZF_OK) { (void)hipFreeAsync(f.s.base, st); f.s.base = nullptr; f.live = false; g_last_error = "bitmap-fed decompress: launch failed"; }
    return rc;
}
// slots_per_cu = 0: as many slots as the device holds (fed_geometry)
int fed_back(FedCall& f, hipStream_t st, uint32_t slots_per_cu) {
    if (!f.live) return LZF_OK;
    lzf::seg_ctx& c = f.s.ctx;
    const uint32_t n = f.n;
    uint8_t* b = static_cast<uint8_t*>(f.s.base);
    // The kernel runs as one wavefront per SLOT — as many as the device holds at once — and the slots share the jobs out in pieces
    // (lz4_decompress_fed.hip): a call with more jobs than slots cuts every job into 16 (measured at 2.2 jobs per slot: 107.4 ms whole, 97.3 / 97.2 / 97.8 / 99.0 / 101.7 ms with
    // 8 / 16 / 32 / 64 / 128 pieces), a smaller one leaves them whole.
    const FedGeometry fg = fed_geometry(st);
    uint32_t slots = fg.slots;
    if (slots_per_cu && slots_per_cu * cu_count() < slots) slots = slots_per_cu * cu_count();
    uint32_t pieces = n > slots && fg.xcc_mask ? 16u : 1u;
    uint32_t pad = 0;
#ifdef LZF_ANALYSIS      // LZF_FED_PIECES = pieces per job (A/B), LZF_FED_SLOTS = slots per CU, LZF_FED_PAD_LDS = bytes of unused LDS per wavefront (residency experiment)
    { static const long e = [] { const char* v = getenv("LZF_FED_PIECES"); return v ? atol(v) : 0L; }(); if (e >= 1 && e <= 4096) pieces = (uint32_t)e; }
    { static const long e = [] { const char* v = getenv("LZF_FED_SLOTS"); return v ? atol(v) : 0L; }(); if (e > 0 && !slots_per_cu) slots = (uint32_t)e * cu_count(); }
    { static const uint32_t e = [] { const char* v = getenv("LZF_FED_PAD_LDS"); return v ? (uint32_t)atol(v) : 0u; }(); pad = e; }
#endif
    if ((uint64_t)n * pieces > 0xFFFFFFF0ull) pieces = 1u;
    if (!fg.xcc_mask) pieces = 1u;
    int rc = LZF_OK;
    lzf::fed_args a{f.jobs, f.res, c.st, c.bits, c.vfrom, f.perm, reinterpret_cast<lzf::fed_state*>(b + f.o_fst), reinterpret_cast<uint32_t*>(b + f.o_tick), fg.xcc_mask ? fg.xcc_mask : 1u, n, c.maxch, pieces, nullptr};
    hipLaunchKernelGGL(lzf::lzf_fed_reset_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_fed32, dim3(slots < n ? slots : n), dim3(64), pad, st, a);
    if (hipGetLastError() != hipSuccess) rc = LZF_E_HIP;
    if (rc == LZF_OK) {
        hipLaunchKernelGGL(k_paired24, dim3(n), dim3(128), 0, st, f.jobs, f.res, n, f.perm, (const lzf::seg_job*)c.st);
        if (hipGetLastError() != hipSuccess) rc = LZF_E_HIP;
    }
    if (hipFreeAsync(f.s.base, st) != hipSuccess && rc == LZF_OK) rc = LZF_E_HIP;
    f.s.base = nullptr; f.live = false;
    if (rc != LZF_OK) g_last_error = "bitmap-fed decompress: launch failed";
    return rc;
}
int fed_decompress(const lzf_decompress_job* d_jobs, lzf_job_result* d_results, uint32_t n, const uint32_t* perm, hipStream_t st, bool* used, uint64_t max_in_hint) {
    *used = false;
    if (n > kFedMaxJobs || max_in_hint <= fed_min_in() || max_in_hint <= fed_min_hint()) return LZF_OK;
    FedCall f;
    int rc = fed_front(f, d_jobs, d_results, n, perm, st, max_in_hint);
    if (rc != LZF_OK || !f.live) return rc;
    *used = true;
    return fed_back(f, st, 0u);
}

#ifdef LZF_ANALYSIS
#include "analysis/capi_analysis.inc"
#endif

}  // namespace

extern "C" {

int lzf_abi_version(void) { return LZFEAR_ABI_VERSION; }
const char* lzf_last_error(void) { return g_last_error.c_str(); }
const char* lzf_last_decompress_launch(void) { return g_last_decompress; }
const char* lzf_last_compress_launch(void) { return g_last_compress; }
int lzf_device_count(void) { return ensure_device(); }

// Launch order (both batch calls): a batch of more jobs than the chip holds at once runs longest job first, or the launch
// ends with a few long jobs running alone.  The order is internal — results[i] always belongs to jobs[i].
int lzf_compress_batch(const lzf_compress_job* d_jobs, lzf_job_result* d_results, uint32_t n_jobs,
                       uint32_t table_kinds, void* hip_stream) {
    if (n_jobs == 0) return LZF_OK;
    if (!d_jobs || !d_results) { g_last_error = "lzf_compress_batch: NULL job/result array"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    keep_pool_memory();
    if ((table_kinds & (LZF_KINDS_U32 | LZF_KINDS_U16)) == 0) table_kinds |= LZF_KINDS_U32 | LZF_KINDS_U16;
    // U32 jobs with a fresh or read-only template table go to the compact-table kernel (18 instead of 10 waves per CU), the
    // others to the general kernel; which is which is in the job array, i.e. in HBM, so both are launched (a wave of the
    // kernel that does not own a job reads the job and returns) unless the caller vouches for the batch with
    // LZF_KINDS_U32_FRESH_ONLY.
    uint32_t use_compact = 1u, use_order = 1u;
    // The latency class: a call with no more jobs than the chip has compute units gives every compact-table job a CU of its own —
    // lzf_compress_team_kernel, three wavefronts per block, input window and table in LDS (lz4_compress_team.inc) — instead of a lone
    // wavefront of the compact kernel.  (Needs a CU's whole LDS; LZF_COMPRESS_TEAM_MAX in the analysis flavour moves the threshold.)
    uint32_t team_max = geometry().lds >= kTeamLds ? kTeamRounds * cu_count() : 0u;
#ifdef LZF_ANALYSIS
    // LZF_COMPRESS_KERNEL = general (everything on lzf_compress_wave_kernel) | compact (no latency class)
    { static const uint32_t which = [] { const char* e = getenv("LZF_COMPRESS_KERNEL"); return !e ? 0u : !strcmp(e, "general") ? 1u : !strcmp(e, "compact") ? 3u : 0u; }();
      static const uint32_t order = analysis_order("LZF_COMPRESS_ORDER");
      use_compact = which == 1u ? 0u : 1u; use_order = order;
      static const long tm = [] { const char* e = getenv("LZF_COMPRESS_TEAM_MAX"); return e ? atol(e) : -1L; }();
      if (tm >= 0 && geometry().lds >= kTeamLds) team_max = (uint32_t)tm;
      if (which != 0u) team_max = 0u; }
#endif
    const bool use_team = use_compact && n_jobs <= team_max;
    const bool fresh_only = use_compact && (table_kinds & LZF_KINDS_U32_FRESH_ONLY);
    uint32_t* perm = nullptr;
    AsyncScratch scratch_owner; scratch_owner.st = st;
    void*& scratch = scratch_owner.p;
    // the cost of a compress job is not known from its size: probe (aux_kernels.hip), then longest first
    // (from four jobs per CU on — not only beyond the 18 per CU the chip holds at once: the order also spreads the expensive blocks over the
    //  compute units of a launch that is resident as a whole; one call over N four-MiB blocks, caller's order against longest first, probe
    //  included: 1 020 blocks 287 / 286 ms, 2 040 338 / 293, 3 060 420 / 305, 3 825 440 / 339, 4 590 488 / 357.  Below one residency only for
    //  calls that vouch for fresh tables — every job then is the probe's kind of job; a call of carried tables is typically thousands of
    //  64 KiB blocks of linked streams, too small to be probed, for which the three extra launches were 15 % of the call: bench config5)
    const bool want_order = use_order && use_compact && (table_kinds & LZF_KINDS_U32) && (use_order == 2u || n_jobs > per_cu(lzf::kCompactLdsBytes) * cu_count() || (fresh_only && n_jobs > 4u * cu_count()));
    uint32_t piece = 65536u, parts = 1u;            // one 64 KiB piece from the middle of each payload (more or smaller pieces order no better)
#ifdef LZF_ANALYSIS      // LZF_PROBE="piece,parts": the cost probe's sample (A/B of the launch order's estimate)
    { static const char* e = getenv("LZF_PROBE"); if (e) { unsigned a = 0, b = 0; if (sscanf(e, "%u,%u", &a, &b) == 2 && a >= 4096u && b >= 1u && b <= 16u) { piece = a; parts = b; } } }
#endif
    const size_t n_probes = want_order ? (size_t)n_jobs * parts : 0u;
    const size_t probes_off = 256;                  // [probe jobs][probe results][perm]
    const size_t res_off = probes_off + align_up(sizeof(lzf_compress_job) * n_probes, 256);
    const size_t perm_off = res_off + align_up(sizeof(lzf_job_result) * n_probes, 256);
    if (want_order) {
        // (the order is an optimisation: without scratch memory the batch simply runs in the caller's order)
        if (hipMallocAsync(&scratch, perm_off + sizeof(uint32_t) * (want_order ? (size_t)n_jobs : 0u), st) != hipSuccess) { (void)hipGetLastError(); scratch = nullptr; }
    }
    if (scratch && want_order) {
        lzf_compress_job* probes = reinterpret_cast<lzf_compress_job*>(static_cast<uint8_t*>(scratch) + probes_off);
        lzf_job_result* pres = reinterpret_cast<lzf_job_result*>(static_cast<uint8_t*>(scratch) + res_off);
        perm = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(scratch) + perm_off);
        LAUNCH(lzf::lzf_cost_probe_jobs_kernel, dim3((uint32_t)((n_probes + 255u) / 256u)), dim3(256), 0, st, d_jobs, probes, n_jobs, piece, parts);
        LAUNCH(k_compact_dry, dim3((uint32_t)n_probes), dim3(64), 0, st, probes, pres, (uint32_t)n_probes, (const uint32_t*)nullptr, 0u);
        LAUNCH(lzf::lzf_order_by_cost_kernel, dim3(1), dim3(1024), 0, st, d_jobs, pres, perm, n_jobs, piece, parts);
    }
    if (table_kinds & LZF_KINDS_U32) {
#ifdef LZF_DBG_DRY_MAIN      // analysis: results[].reserved = probe batches + sequences of the whole job (no output)
        if (use_compact) LAUNCH(k_compact_dry, dim3(n_jobs), dim3(64), 0, st, d_jobs, d_results, n_jobs, (const uint32_t*)perm, 0u);
#else
        if (use_team) {
            LAUNCH(lzf::lzf_compress_team_kernel, dim3(n_jobs), dim3(192), 0, st, d_jobs, d_results, n_jobs, (const uint32_t*)perm, fresh_only ? 1u : 0u);
            // caller-owned tables (linked streams): the same team, compiled with the table carry; it leaves the compact jobs to the kernel above
            if (!fresh_only) LAUNCH(lzf::lzf_compress_team_carry_kernel, dim3(n_jobs), dim3(192), 0, st, d_jobs, d_results, n_jobs, (const uint32_t*)perm);
        }
        else if (use_compact) {
            uint32_t pad = 0;
#ifdef LZF_ANALYSIS      // LZF_COMPACT_PAD_LDS = bytes of unused LDS per wavefront: fewer resident waves per CU (the residency experiment)
            { static const uint32_t e = [] { const char* v = getenv("LZF_COMPACT_PAD_LDS"); return v ? (uint32_t)atol(v) : 0u; }(); pad = e; }
#endif
            LAUNCH(k_compact, dim3(n_jobs), dim3(64), pad, st, d_jobs, d_results, n_jobs, (const uint32_t*)perm, fresh_only ? 1u : 0u);
        }
#endif
        // (behind the team kernel the general kernel skips what that one takes — compact jobs and caller-owned U32 tables — unless a job was handed back)
        if (!fresh_only) LAUNCH(k_general_u32, dim3(n_jobs), dim3(64), 0, st, d_jobs, d_results, n_jobs, use_team ? 2u : use_compact, (const uint32_t*)perm);
    }
    if (table_kinds & LZF_KINDS_U16)
        LAUNCH(k_general_u16, dim3(n_jobs), dim3(64), 0, st, d_jobs, d_results, n_jobs, 0u, (const uint32_t*)perm);
    g_last_compress = !(table_kinds & LZF_KINDS_U32) ? "lzf_compress_wave_kernel<U16>"
                    : !use_compact ? "lzf_compress_wave_kernel (analysis: general)"
                    : use_team ? (fresh_only ? "lzf_compress_team_kernel" : "lzf_compress_team_kernel + lzf_compress_team_carry_kernel + lzf_compress_wave_kernel")
                               : (fresh_only ? "lzf_compress_compact_kernel" : "lzf_compress_compact_kernel + lzf_compress_wave_kernel");
    HIP_TRY(scratch_owner.release());
    return LZF_OK;
}

int lzf_decompress_batch(const lzf_decompress_job* d_jobs, lzf_job_result* d_results, uint32_t n_jobs, void* hip_stream) {
    return lzf_decompress_batch_sized(d_jobs, d_results, n_jobs, ~0ull, hip_stream);
}

int lzf_decompress_batch_sized(const lzf_decompress_job* d_jobs, lzf_job_result* d_results, uint32_t n_jobs, uint64_t max_input_len, void* hip_stream) {
    if (n_jobs == 0) return LZF_OK;
    if (!d_jobs || !d_results) { g_last_error = "lzf_decompress_batch: NULL job/result array"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    keep_pool_memory();
    uint32_t use_order = 1u;
    bool perm_ok = true;
#ifdef LZF_ANALYSIS
    const int forced = analysis_decompress_variant();
    { static const uint32_t order = analysis_order("LZF_DECOMPRESS_ORDER"); use_order = order; }
    if (forced != kVariantAuto) perm_ok = analysis_perm_ok(forced);
#endif
    // more blocks than the chip holds at once: longest (most compressed bytes) first
    AsyncScratch perm_owner; perm_owner.st = st;
    uint32_t* perm = nullptr;
    if (perm_ok && (use_order == 2u || (use_order == 1u && n_jobs > per_cu(20u * 1024u) * cu_count()))) {
        // perm[n] + est[n]: the jobs by their estimated cost, largest first
        if (hipMallocAsync(&perm_owner.p, 2u * sizeof(uint32_t) * (size_t)n_jobs, st) != hipSuccess) { (void)hipGetLastError(); perm_owner.p = nullptr; }   // (then: the caller's order)
        perm = static_cast<uint32_t*>(perm_owner.p);
        if (perm) {
            uint32_t* const est = perm + n_jobs;
            uint32_t len_shift = 2u;       // a job's cost: its sequences + a quarter of its compressed bytes (measured: 403 GiB/s with the sequences alone, 414-418 with len >> 4 .. len >> 1)
#ifdef LZF_ANALYSIS      // LZF_ORDER_LEN_SHIFT=k: the estimate + input length >> k (A/B of the cost proxy)
            { static const uint32_t k = [] { const char* e = getenv("LZF_ORDER_LEN_SHIFT"); return e ? (uint32_t)atol(e) : 2u; }(); len_shift = k; }
#endif
            LAUNCH(lzf::lzf_decompress_cost_kernel, dim3(n_jobs), dim3(64), 0, st, d_jobs, n_jobs, est, len_shift);
            LAUNCH(lzf::lzf_order_by_estimate_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t*)est, perm, n_jobs);
        }
    }
    const uint32_t* cperm = perm;
#ifdef LZF_ANALYSIS
    if (forced != kVariantAuto) {
        g_last_decompress = "analysis variant (LZF_DECOMPRESS_KERNEL)";
        rc = analysis_launch_decompress(forced, d_jobs, d_results, n_jobs, cperm, st);
        HIP_TRY(perm_owner.release());
        return rc;
    }
#endif
    // Batches that leave the chip mostly empty with one workgroup per block: the segmented pipeline (a block decoded by many
    // wavefronts), then the pair kernel over the jobs it left (prefix / existing output, errors, sizes outside its window).
    uint32_t seg_min_in = kSegMinIn; bool seg_on = n_jobs <= seg_max_jobs();
#ifdef LZF_ANALYSIS
    { static const int mode = [] { const char* e = getenv("LZF_DECOMPRESS_KERNEL"); return !e ? 0 : !strcmp(e, "seg") ? 1 : (!strcmp(e, "noseg") || !strcmp(e, "fed")) ? 2 : 0; }();
      static const uint32_t min_in = [] { const char* e = getenv("LZF_SEG_MIN_IN"); return e ? (uint32_t)atol(e) : 0u; }();
      if (mode == 1) { seg_on = true; seg_min_in = min_in; }
      if (mode == 2) seg_on = false; }
#endif
    if (seg_on) {
        bool used = false; uint32_t ring = 0;
        rc = seg_decompress(d_jobs, d_results, n_jobs, seg_min_in, st, &used, max_input_len, &ring);
        if (used) g_last_decompress = ring == 131072u ? "segmented: lzf_seg_resolve_pair_kernel<131072> + lzf_decompress_paired_kernel<4096,48,640>"      // (the ring the call really used)
                                     : ring == 65536u ? "segmented: lzf_seg_resolve_pair_kernel<65536> + lzf_decompress_paired_kernel<4096,48,640>"
                                                      : "segmented: lzf_seg_resolve_pair_kernel<32768> + lzf_decompress_paired_kernel<4096,48,640>";
        if (rc != LZF_OK || used) { HIP_TRY(perm_owner.release()); return rc; }
    }
    // The producer/consumer pair kernel, with 48-byte regions while every block's workgroup is resident at once (lowest
    // latency per block: the copy stage is the critical path, the parse rides along) and 24-byte regions beyond that
    // (smaller LDS footprint, more blocks in flight); batches of more than eight times that many blocks (small blocks,
    // typically) go to the one-wave staged16 kernel, which has no per-block pipeline to fill.
    const uint32_t resident48 = per_cu(20u * 1024u) * cu_count();      // workgroups of the 48-byte form one device holds (20 KB of LDS each: 8 per CU on MI355X)
    // the bitmap-fed path beyond what the 24-byte pair kernel holds at once (12.5 KiB of LDS per pair: 13 per CU, 3 328 on MI355X): up to
    // there every block has its pair of wavefronts for itself and the pair kernel is quicker (2 107 jobs 30.5 against 36.0 ms, 3 038 jobs 34.0
    // against 38.0; 4 214 jobs 51.1 against 46.1, 8 085 jobs 81.6 against 73.5: profiles/r06_fed_kernel_study.txt)
    const uint32_t resident24 = per_cu(12800u) * cu_count();
    bool fed_on = n_jobs > resident24;
#ifdef LZF_ANALYSIS
    { static const int mode = [] { const char* e = getenv("LZF_DECOMPRESS_KERNEL"); return !e ? 0 : !strcmp(e, "fed") ? 1 : !strcmp(e, "nofed") ? 2 : 0; }();
      if (mode == 1) fed_on = true;
      if (mode == 2) fed_on = false; }
#endif
    if (fed_on) {
        bool used = false;
        rc = fed_decompress(d_jobs, d_results, n_jobs, cperm, st, &used, max_input_len);
        if (used) g_last_decompress = "bitmap-fed: lzf_seg_parse_kernel + lzf_decompress_fed_kernel<4096,32,352> + lzf_decompress_paired_kernel<4096,24,384>";
        if (rc != LZF_OK || used) { HIP_TRY(perm_owner.release()); return rc; }
    }
    g_last_decompress = n_jobs <= resident48 ? "lzf_decompress_paired_kernel<4096,48,640>" : n_jobs <= 8u * resident48 ? "lzf_decompress_paired_kernel<4096,24,384>"
                                                                                                             : "lzf_decompress_batched_kernel<4096,16,256,staged>";
    if (n_jobs <= resident48)
        LAUNCH(k_paired48, dim3(n_jobs), dim3(128), 0, st, d_jobs, d_results, n_jobs, cperm, (const lzf::seg_job*)nullptr);
    else if (n_jobs <= 8u * resident48)
        LAUNCH(k_paired24, dim3(n_jobs), dim3(128), 0, st, d_jobs, d_results, n_jobs, cperm, (const lzf::seg_job*)nullptr);
    else
        LAUNCH(k_staged16, dim3(n_jobs), dim3(64), 0, st, d_jobs, d_results, n_jobs, cperm);
    HIP_TRY(perm_owner.release());
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
    // up to a few thousand buffers (block / content checksums of large blocks): one wave per buffer, streaming loads; beyond
    // that (many small blocks) 16 hashes per wave keep more chains per CU
    if (n <= 8192u) hipLaunchKernelGGL(lzf::lzf_xxh32_wave_kernel, dim3(n), dim3(64), 0, static_cast<hipStream_t>(hip_stream), d_ptrs, d_lens, d_out, n);
    else hipLaunchKernelGGL(lzf::lzf_xxh32_kernel, dim3((n + 15) / 16), dim3(64), 0, static_cast<hipStream_t>(hip_stream), d_ptrs, d_lens, d_out, n);
    HIP_TRY(hipGetLastError());
    return LZF_OK;
}

#ifdef LZF_ANALYSIS
// Analysis only: run the segmented pipeline up to a stage and copy its scratch areas to host buffers (NULL = skip).
// geom[0..3] = maxch, maxtile, chunk words, records in the arena.
int lzf_debug_seg(const lzf_decompress_job* d_jobs, lzf_job_result* d_results, uint32_t n, uint32_t min_in, uint32_t upto,
                  void* h_state, void* h_bits, void* h_xexit, void* h_vfrom, void* h_tile_tok, void* h_tile_out, void* h_recs, uint64_t recs_bytes,
                  uint32_t* geom) {
    int rc = ensure_device();
    if (rc < 0) return rc;
    SegScratch s;
    if (!seg_alloc(s, d_jobs, d_results, n, min_in, nullptr)) return LZF_E_HIP;
    const lzf::seg_ctx& c = s.ctx;
    rc = seg_launch(c, upto, nullptr);
    HIP_TRY(hipDeviceSynchronize());
    if (geom) { geom[0] = c.maxch; geom[1] = c.maxtile; geom[2] = lzf::kSegChunkWords; geom[3] = (uint32_t)c.rec_cap; }
    if (h_state) HIP_TRY(hipMemcpy(h_state, c.st, sizeof(lzf::seg_job) * n, hipMemcpyDeviceToHost));
    if (h_bits) HIP_TRY(hipMemcpy(h_bits, c.bits, sizeof(uint32_t) * (size_t)n * c.maxch * lzf::kSegChunkWords, hipMemcpyDeviceToHost));
    if (h_xexit) HIP_TRY(hipMemcpy(h_xexit, c.xexit, sizeof(uint32_t) * (size_t)n * c.maxch, hipMemcpyDeviceToHost));
    if (h_vfrom) HIP_TRY(hipMemcpy(h_vfrom, c.vfrom, sizeof(uint32_t) * (size_t)n * c.maxch, hipMemcpyDeviceToHost));
    if (h_tile_tok) HIP_TRY(hipMemcpy(h_tile_tok, c.tile_tok, sizeof(uint32_t) * (size_t)n * c.maxtile, hipMemcpyDeviceToHost));
    if (h_tile_out) HIP_TRY(hipMemcpy(h_tile_out, c.tile_out, sizeof(uint32_t) * (size_t)n * c.maxtile, hipMemcpyDeviceToHost));
    if (h_recs) { uint64_t nb = sizeof(lzf::u32x4) * c.rec_cap; if (nb > recs_bytes) nb = recs_bytes; HIP_TRY(hipMemcpy(h_recs, c.recs, nb, hipMemcpyDeviceToHost)); }
    HIP_TRY(hipFree(s.base));
    return rc;
}
#endif

// ---------------------------------------------------------------------------------------
// host-buffer helpers: stage -> launch -> copy back.  Synchronous.  Every job's bytes travel through the pinned slab of
// host_staging.h in 4 MiB pieces (worker threads memcpy, one asynchronous DMA per piece), device scratch is kept
// between calls; the only per-job host work is the layout arithmetic.
// ---------------------------------------------------------------------------------------
int lzf_compress_batch_host(const lzf_compress_job* jobs, lzf_job_result* results, uint32_t n_jobs) {
    if (n_jobs == 0) return LZF_OK;
    if (!jobs || !results) { g_last_error = "lzf_compress_batch_host: NULL argument"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    using lzf_host::Seg; using lzf_host::Staging;
    // layout: one slab [inputs | tables] going up, one slab [outputs] coming back
    std::vector<size_t> in_off(n_jobs), out_off(n_jobs), tab_off(n_jobs);
    size_t in_total = 0, out_total = 0;
    uint32_t kinds = 0;
    std::vector<Seg> up;
    static_assert(sizeof(lzf_u32_table) == sizeof(lzf_u16_table), "table structs share a slab slot size");
    for (uint32_t i = 0; i < n_jobs; ++i) {
        if (jobs[i].table_kind > LZF_TABLE_U16) { g_last_error = "bad table_kind"; return LZF_E_INVALID; }
        kinds |= jobs[i].table_kind == LZF_TABLE_U32 ? LZF_KINDS_U32 : LZF_KINDS_U16;
        in_off[i] = in_total; in_total = align_up(in_total + jobs[i].input_len, 256);
        if (jobs[i].input_len) up.push_back({in_off[i], const_cast<uint8_t*>(jobs[i].input), (size_t)jobs[i].input_len});
    }
    for (uint32_t i = 0; i < n_jobs; ++i) {
        tab_off[i] = in_total;
        if (jobs[i].table) { up.push_back({tab_off[i], static_cast<uint8_t*>(jobs[i].table), sizeof(lzf_u32_table)}); in_total = align_up(in_total + sizeof(lzf_u32_table), 256); }
    }
    for (uint32_t i = 0; i < n_jobs; ++i) { out_off[i] = out_total; out_total = align_up(out_total + jobs[i].out_cap, 256); }
    Staging& sg = Staging::get();
    std::lock_guard<std::mutex> guard(sg.lock());
    hipStream_t cs = sg.stream(0);
    uint8_t* const din = static_cast<uint8_t*>(sg.device(0, in_total));
    uint8_t* const dout = static_cast<uint8_t*>(sg.device(1, out_total));
    lzf_compress_job* const djobs = static_cast<lzf_compress_job*>(sg.device(3, sizeof(lzf_compress_job) * n_jobs));
    lzf_job_result* const dres = static_cast<lzf_job_result*>(sg.device(4, sizeof(lzf_job_result) * n_jobs));
    if (!cs || !din || !dout || !djobs || !dres || !sg.pinned(in_total > out_total ? in_total : out_total)) return fail_hip(hipErrorOutOfMemory, "staging memory");
    std::vector<lzf_compress_job> dj(jobs, jobs + n_jobs);
    for (uint32_t i = 0; i < n_jobs; ++i) {
        dj[i].input = din + in_off[i];
        dj[i].out = dout + out_off[i];
        if (jobs[i].table) dj[i].table = din + tab_off[i];
    }
    HIP_TRY(sg.upload(up, in_total, din));
    HIP_TRY(hipMemcpyAsync(djobs, dj.data(), sizeof(lzf_compress_job) * n_jobs, hipMemcpyHostToDevice, cs));
    HIP_TRY(sg.join_copies(cs));
    rc = lzf_compress_batch(djobs, dres, n_jobs, kinds, cs);
    if (rc != LZF_OK) { (void)hipDeviceSynchronize(); return rc; }
    HIP_TRY(hipMemcpyAsync(results, dres, sizeof(lzf_job_result) * n_jobs, hipMemcpyDeviceToHost, cs));
    HIP_TRY(hipStreamSynchronize(cs));
    std::vector<Seg> down;
    for (uint32_t i = 0; i < n_jobs; ++i) {
        if (results[i].status == LZF_OK && results[i].out_len) down.push_back({out_off[i], jobs[i].out, (size_t)results[i].out_len});
        if (jobs[i].table && !(jobs[i].flags & LZF_CJOB_TABLE_READONLY) && results[i].status != LZF_CONTRACT)
            HIP_TRY(hipMemcpyAsync(jobs[i].table, din + tab_off[i], sizeof(lzf_u32_table), hipMemcpyDeviceToHost, cs));
    }
    HIP_TRY(sg.download(down, out_total, dout, cs));
    HIP_TRY(hipStreamSynchronize(cs));
    return LZF_OK;
}

int lzf_decompress_batch_host(const lzf_decompress_job* jobs, lzf_job_result* results, uint32_t n_jobs) {
    if (n_jobs == 0) return LZF_OK;
    if (!jobs || !results) { g_last_error = "lzf_decompress_batch_host: NULL argument"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    using lzf_host::Seg; using lzf_host::Staging;
    // one slab going up: [inputs | prefixes | existing output]; the output slab is laid out the same way for the way back
    std::vector<size_t> in_off(n_jobs), pre_off(n_jobs), out_off(n_jobs);
    size_t in_total = 0, out_total = 0;
    std::vector<Seg> up, up_out;
    for (uint32_t i = 0; i < n_jobs; ++i) {
        if (jobs[i].out_existing_len > jobs[i].out_cap) { g_last_error = "out_existing_len > out_cap"; return LZF_E_INVALID; }
        in_off[i] = in_total; in_total = align_up(in_total + jobs[i].input_len, 256);
        if (jobs[i].input_len) up.push_back({in_off[i], const_cast<uint8_t*>(jobs[i].input), (size_t)jobs[i].input_len});
    }
    for (uint32_t i = 0; i < n_jobs; ++i) {
        pre_off[i] = in_total; in_total = align_up(in_total + jobs[i].prefix_len, 256);
        if (jobs[i].prefix_len) up.push_back({pre_off[i], const_cast<uint8_t*>(jobs[i].prefix), (size_t)jobs[i].prefix_len});
    }
    for (uint32_t i = 0; i < n_jobs; ++i) {
        out_off[i] = out_total; out_total = align_up(out_total + jobs[i].out_cap, 256);
        if (jobs[i].out_existing_len) up_out.push_back({out_off[i], jobs[i].out, (size_t)jobs[i].out_existing_len});
    }
    Staging& sg = Staging::get();
    std::lock_guard<std::mutex> guard(sg.lock());
    hipStream_t cs = sg.stream(0);
    uint8_t* const din = static_cast<uint8_t*>(sg.device(0, in_total));
    uint8_t* const dout = static_cast<uint8_t*>(sg.device(1, out_total));
    lzf_decompress_job* const djobs = static_cast<lzf_decompress_job*>(sg.device(3, sizeof(lzf_decompress_job) * n_jobs));
    lzf_job_result* const dres = static_cast<lzf_job_result*>(sg.device(4, sizeof(lzf_job_result) * n_jobs));
    if (!cs || !din || !dout || !djobs || !dres || !sg.pinned(in_total > out_total ? in_total : out_total)) return fail_hip(hipErrorOutOfMemory, "staging memory");
    std::vector<lzf_decompress_job> dj(jobs, jobs + n_jobs);
    for (uint32_t i = 0; i < n_jobs; ++i) {
        dj[i].input = din + in_off[i];
        dj[i].prefix = din + pre_off[i];
        dj[i].out = dout + out_off[i];
    }
    if (!up_out.empty()) {                          // (the slab is used for one move at a time: existing output first, and wait for it)
        HIP_TRY(sg.upload(up_out, out_total, dout));
        HIP_TRY(sg.join_copies(cs));
        HIP_TRY(hipStreamSynchronize(cs));
    }
    HIP_TRY(sg.upload(up, in_total, din));
    HIP_TRY(hipMemcpyAsync(djobs, dj.data(), sizeof(lzf_decompress_job) * n_jobs, hipMemcpyHostToDevice, cs));
    HIP_TRY(sg.join_copies(cs));
    uint64_t max_in = 0;
    for (uint32_t i = 0; i < n_jobs; ++i) if (jobs[i].input_len > max_in) max_in = jobs[i].input_len;
    rc = lzf_decompress_batch_sized(djobs, dres, n_jobs, max_in, cs);
    if (rc != LZF_OK) { (void)hipDeviceSynchronize(); return rc; }
    HIP_TRY(hipMemcpyAsync(results, dres, sizeof(lzf_job_result) * n_jobs, hipMemcpyDeviceToHost, cs));
    HIP_TRY(hipStreamSynchronize(cs));
    std::vector<Seg> down;
    for (uint32_t i = 0; i < n_jobs; ++i) {
        uint64_t n = results[i].out_len;
        if (n > jobs[i].out_cap) n = jobs[i].out_cap;
        if (n > jobs[i].out_existing_len)
            down.push_back({out_off[i] + (size_t)jobs[i].out_existing_len, jobs[i].out + jobs[i].out_existing_len, (size_t)(n - jobs[i].out_existing_len)});
    }
    HIP_TRY(sg.download(down, out_total, dout, cs));
    return LZF_OK;
}

// ---- EncoderTable on host tables (src/raw/compress/mod.rs:40-61, :64-74, :88-99) --------------------------------------
int lzf_table_replace_host(void* table, uint32_t table_kind, const uint8_t* input, uint64_t input_len, uint64_t pos, uint64_t* previous) {
    if (!table || (!input && input_len) || table_kind > LZF_TABLE_U16) { g_last_error = "lzf_table_replace_host: bad argument"; return LZF_E_INVALID; }
    if (pos > input_len) return LZF_CONTRACT;                         // &input[offset..] panics
    const uint64_t rem = input_len - pos;
    if (table_kind == LZF_TABLE_U32) {
        lzf_u32_table* t = static_cast<lzf_u32_table*>(table);
        const uint64_t o = pos + t->offset;
        if (o > 0xFFFFFFFFull || o < pos) return LZF_CONTRACT;         // :67 try_into().expect
        uint64_t v = 0;
        if (rem >= 8) memcpy(&v, input + pos, 8);                      // :43 input.get(..8) or 0 (little-endian host = little-endian GPU)
        const uint32_t slot = (uint32_t)(((v << 24) * 889523592379ull) >> 52);
        const uint32_t old = t->dict[slot];
        t->dict[slot] = (uint32_t)o;
        if (previous) *previous = old > t->offset ? old - t->offset : 0;
    } else {
        lzf_u16_table* t = static_cast<lzf_u16_table*>(table);
        const uint64_t o = pos + t->offset;
        if (o > 0xFFFFull || o < pos) return LZF_CONTRACT;             // :92
        if (rem < 4) return LZF_CONTRACT;                               // :59 read_u32 on a short slice panics
        uint32_t v; memcpy(&v, input + pos, 4);
        const uint32_t slot = (uint32_t)(v * 2654435761u) >> 19;
        const uint16_t old = t->dict[slot];
        t->dict[slot] = (uint16_t)o;
        if (previous) *previous = old > t->offset ? old - t->offset : 0;
    }
    return LZF_OK;
}

int lzf_table_offset_host(void* table, uint32_t table_kind, uint64_t add) {
    if (!table || table_kind > LZF_TABLE_U16) { g_last_error = "lzf_table_offset_host: bad argument"; return LZF_E_INVALID; }
    if (table_kind == LZF_TABLE_U32) static_cast<lzf_u32_table*>(table)->offset += add;
    else static_cast<lzf_u16_table*>(table)->offset += add;
    return LZF_OK;
}

// ---- compress2 for any writer: device compress against the worst-case bound, then the reference's write calls replayed ----
int lzf_compress2_host_writer(const uint8_t* input, uint64_t input_len, uint64_t cursor, void* table, uint32_t table_kind,
                              lzf_write_all_fn write_all, void* ctx, int* writer_error) {
    if ((!input && input_len) || !write_all || table_kind > LZF_TABLE_U16) { g_last_error = "lzf_compress2_host_writer: bad argument"; return LZF_E_INVALID; }
    if (writer_error) *writer_error = 0;
    const uint64_t payload = cursor < input_len ? input_len - cursor : 0;
    const uint64_t bound = payload + payload / 255 + 16;
    std::vector<uint8_t> out(bound);
    static_assert(sizeof(lzf_u32_table) == sizeof(lzf_u16_table), "one scratch copy serves both table kinds");
    std::vector<uint8_t> t0(sizeof(lzf_u32_table), 0), t1;
    if (table) memcpy(t0.data(), table, t0.size());                    // (the state on entry: a refused write needs a second run from it)
    t1 = t0;
    lzf_compress_job job{};
    job.input = input; job.input_len = input_len; job.cursor = cursor;
    job.out = out.data(); job.out_cap = bound; job.table = table ? t1.data() : nullptr; job.table_kind = table_kind;
    lzf_job_result res{};
    int rc = lzf_compress_batch_host(&job, &res, 1);
    if (rc != LZF_OK) return rc;
    if (res.status != LZF_OK) return res.status;                       // LZF_CONTRACT (the bound cannot be exceeded)
    const uint8_t* p = out.data();
    const uint8_t* const end = p + res.out_len;
    // one call of the writer; false = refused
    int werr = 0;
    auto put = [&](const uint8_t* d, size_t n) -> bool { if (n == 0) return true; werr = write_all(ctx, d, n); return werr == 0; };
    // the tail of a length (mod.rs:243-260) as it sits in the stream at q: k 0xFF bytes and the remainder byte
    auto put_tail = [&](const uint8_t*& q) -> bool {
        size_t k = 0; while (q[k] == 0xFF) ++k;
        for (size_t i = 0; i < k / 4; ++i) { if (!put(q, 4)) return false; q += 4; }
        for (size_t i = 0; i < k % 4; ++i) { if (!put(q, 1)) return false; q += 1; }
        if (!put(q, 1)) return false;
        q += 1;
        return true;
    };
    bool refused = false;
    const uint8_t* group = p;
    while (p < end && !refused) {
        group = p;
        const uint8_t tok = *p;
        const uint8_t* q = p + 1;
        if (!put(p, 1)) { refused = true; break; }                      // writer.write_u8(token)
        size_t L = tok >> 4;
        if (L == 15) { const uint8_t* t = q; while (*t == 0xFF) { L += 255; ++t; } L += *t; if (!put_tail(q)) { refused = true; break; } }
        if (!put(q, L)) { refused = true; break; }                      // writer.write_all(literal)
        q += L;
        if (q >= end) { p = q; break; }                                 // the literal-only section that ends the block (:182-189)
        if (!put(q, 2)) { refused = true; break; }                      // write_u16::<LE>(offset)
        q += 2;
        if ((tok & 15) == 15) { if (!put_tail(q)) { refused = true; break; } }
        p = q;
    }
    if (!refused) {
        if (table) memcpy(table, t1.data(), t1.size());
        return LZF_OK;
    }
    if (writer_error) *writer_error = werr;
    if (table) {
        // the table as the reference leaves it: after the search of the refused sequence, i.e. compress2 into a sink that
        // takes everything in front of that sequence and not the sequence itself
        const uint8_t* q = group; const uint8_t tok = *q++; size_t L = tok >> 4;
        if (L == 15) { while (*q == 0xFF) { L += 255; ++q; } L += *q++; }
        q += L;
        if (q < end) { q += 2; if ((tok & 15) == 15) { while (*q == 0xFF) ++q; ++q; } }
        const uint64_t gsize = (uint64_t)(q - group);
        t1 = t0;
        job.out_cap = (uint64_t)(group - out.data()) + gsize - 1;
        job.table = t1.data();
        rc = lzf_compress_batch_host(&job, &res, 1);
        if (rc != LZF_OK) return rc;
        memcpy(table, t1.data(), t1.size());
    }
    return LZF_OUTPUT_FULL;
}

int lzf_xxh32_batch_host(const uint8_t* const* ptrs, const uint64_t* lens, uint32_t* out, uint32_t n) {
    if (n == 0) return LZF_OK;
    if (!ptrs || !lens || !out) { g_last_error = "lzf_xxh32_batch_host: NULL argument"; return LZF_E_INVALID; }
    int rc = ensure_device();
    if (rc < 0) return rc;
    using lzf_host::Seg; using lzf_host::Staging;
    std::vector<Seg> up; std::vector<const uint8_t*> dp(n);
    size_t total = 0;
    for (uint32_t i = 0; i < n; ++i) { if (lens[i]) up.push_back({total, const_cast<uint8_t*>(ptrs[i]), (size_t)lens[i]}); dp[i] = reinterpret_cast<const uint8_t*>(total); total = align_up(total + lens[i], 256); }
    Staging& sg = Staging::get();
    std::lock_guard<std::mutex> guard(sg.lock());
    hipStream_t cs = sg.stream(0);
    uint8_t* const din = static_cast<uint8_t*>(sg.device(0, total));
    const size_t lp = 0, ll = align_up(sizeof(void*) * n, 256), lo = ll + align_up(sizeof(uint64_t) * n, 256);
    uint8_t* const dl = static_cast<uint8_t*>(sg.device(5, lo + sizeof(uint32_t) * n));
    if (!cs || !din || !dl || !sg.pinned(total)) return fail_hip(hipErrorOutOfMemory, "staging memory");
    for (uint32_t i = 0; i < n; ++i) dp[i] = din + reinterpret_cast<size_t>(dp[i]);
    HIP_TRY(sg.upload(up, total, din));
    HIP_TRY(hipMemcpyAsync(dl + lp, dp.data(), sizeof(void*) * n, hipMemcpyHostToDevice, cs));
    HIP_TRY(hipMemcpyAsync(dl + ll, lens, sizeof(uint64_t) * n, hipMemcpyHostToDevice, cs));
    HIP_TRY(sg.join_copies(cs));
    rc = lzf_xxh32_batch(reinterpret_cast<const uint8_t* const*>(dl + lp), reinterpret_cast<const uint64_t*>(dl + ll), reinterpret_cast<uint32_t*>(dl + lo), n, cs);
    if (rc != LZF_OK) { (void)hipDeviceSynchronize(); return rc; }
    HIP_TRY(hipMemcpyAsync(out, dl + lo, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, cs));
    HIP_TRY(hipStreamSynchronize(cs));
    return LZF_OK;
}

}  // extern "C"
