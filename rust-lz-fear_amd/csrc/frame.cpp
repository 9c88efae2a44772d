// frame.cpp — host-side LZ4 frame layer over the GPU block codec (include/lzfear_frame.h).
//
// Mirrors lz-fear's src/framed/{compress,decompress,header}.rs; every block goes through the HIP
// kernels via lzf_compress_batch_host / lzf_decompress_batch_host — there is no CPU codec here.
// All blocks of all frames of a call go into the same launches (independent-block frames: one launch; linked-block
// frames — table and 64 KiB window carry, compress.rs:271-275, decompress.rs:253-269 — one launch per block index, every
// stream of the call advancing together); the one-frame entry points are batches of one.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include "../../include/lzfear_frame.h"
#include "host_staging.h"

namespace {
// a HIP failure inside a driver: nothing asynchronous may still read the caller's (or this call's) host arrays when it returns
inline int fail_hip() { (void)hipDeviceSynchronize(); return LZF_E_HIP; }

inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

// ---- XXH32 (streaming), as used by the frame format for header / block / content checksums
constexpr uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
struct Xxh32 {
    uint32_t v[4]; uint8_t buf[16]; uint32_t fill = 0; uint64_t total = 0; uint32_t seed;
    explicit Xxh32(uint32_t s = 0) : seed(s) { v[0] = s + P1 + P2; v[1] = s + P2; v[2] = s; v[3] = s - P1; }
    void stripe(const uint8_t* p) { for (int i = 0; i < 4; ++i) v[i] = rotl(v[i] + rd32(p + 4 * i) * P2, 13) * P1; }
    void update(const uint8_t* p, size_t n) {
        total += n;
        if (fill) {
            size_t take = 16 - fill; if (take > n) take = n;
            memcpy(buf + fill, p, take); fill += (uint32_t)take; p += take; n -= take;
            if (fill < 16) return;
            stripe(buf); fill = 0;
        }
        for (; n >= 16; p += 16, n -= 16) stripe(p);
        if (n) { memcpy(buf, p, n); fill = (uint32_t)n; }
    }
    uint32_t digest() const {
        uint32_t h = total >= 16 ? rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18) : seed + P5;
        h += (uint32_t)total;
        const uint8_t* p = buf; uint32_t n = fill;
        for (; n >= 4; p += 4, n -= 4) h = rotl(h + rd32(p) * P3, 17) * P4;
        for (; n; ++p, --n) h = rotl(h + (*p) * P5, 11) * P1;
        h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
        return h;
    }
};

// header.rs:8-16
constexpr uint8_t FL_INDEP = 0x20, FL_BLOCKSUM = 0x10, FL_CSIZE = 0x08, FL_CSUM = 0x04, FL_DICTID = 0x01;
constexpr uint32_t INCOMPRESSIBLE = 0x80000000u;   // framed/mod.rs:18

// header.rs:53-62 BlockDescriptor::new
int bd_new(uint64_t maxsize, uint8_t* bd) {
    unsigned tz = maxsize ? (unsigned)__builtin_ctzll(maxsize) : 64;
    unsigned maybe = ((tz > 8 ? tz - 8 : 0) / 2) & 0xFF;
    uint8_t b = (uint8_t)(maybe << 4);
    if (b & 0x8F) return LZF_F_PANIC;                           // :55 parse(..).unwrap()
    unsigned size = (b >> 4) & 7;
    if (size < 4 || (1ull << (size * 2 + 8)) != maxsize) return LZF_F_INVALID_BLOCK_SIZE;
    *bd = b;
    return LZF_OK;
}

// compress.rs:163-200: magic, FLG, BD, [content size], [dict id], HC
size_t write_header(const lzf_settings* s, uint8_t bd, uint8_t* out) {
    uint8_t flags = 0;
    if (s->independent_blocks) flags |= FL_INDEP;
    if (s->block_checksums) flags |= FL_BLOCKSUM;
    if (s->content_checksum) flags |= FL_CSUM;
    if (s->has_dictionary_id) flags |= FL_DICTID;
    if (s->has_content_size) flags |= FL_CSIZE;
    size_t w = 0;
    wr32(out, LZF_MAGIC); w += 4;
    out[w++] = (uint8_t)((1 << 6) | flags);
    out[w++] = bd;
    if (s->has_content_size) { wr32(out + w, (uint32_t)s->content_size); wr32(out + w + 4, (uint32_t)(s->content_size >> 32)); w += 8; }
    if (s->has_dictionary_id) { wr32(out + w, s->dictionary_id); w += 4; }
    Xxh32 h; h.update(out + 4, w - 4);
    out[w++] = (uint8_t)(h.digest() >> 8);
    return w;
}

// Template table of compress.rs:202-214 built by the GPU seeding kernel.
int seeded_template(const uint8_t* dict, size_t dict_len, lzf_u32_table* host_table) {
    memset(host_table, 0, sizeof *host_table);
    if (!dict || dict_len < 8) return LZF_OK;
    void *d_dict = nullptr, *d_tab = nullptr;
    if (hipMalloc(&d_dict, dict_len) != hipSuccess || hipMalloc(&d_tab, sizeof(lzf_u32_table)) != hipSuccess) {
        if (d_dict) (void)hipFree(d_dict);
        return fail_hip();
    }
    int rc = LZF_OK;
    if (hipMemcpy(d_dict, dict, dict_len, hipMemcpyHostToDevice) != hipSuccess) rc = LZF_E_HIP;
    if (rc == LZF_OK) rc = lzf_table_seed_from_dictionary((lzf_u32_table*)d_tab, (const uint8_t*)d_dict, dict_len, nullptr);
    if (rc == LZF_OK && hipMemcpy(host_table, d_tab, sizeof *host_table, hipMemcpyDeviceToHost) != hipSuccess) rc = LZF_E_HIP;
    (void)hipFree(d_dict); (void)hipFree(d_tab);
    return rc;
}

std::atomic<uint64_t> g_host_block_hashes{0}, g_reader_device_hashes{0};
size_t g_budget = 0;     // lzf_frame_set_memory_budget (0: half of the free device memory)

// One block of a frame as the scan finds it (decompress.rs:217-235).
struct Blk { const uint8_t* data; uint32_t len; bool compressed; uint32_t want_sum; size_t end_off; };   // want_sum: the checksum behind it; end_off: input read once it is
struct FrameScan {
    size_t consumed = 0;          // bytes of the input read
    int err = LZF_OK;             // structural error that ends the scan (reported in stream order)
    bool endmark = false;
    uint32_t want_content = 0;    // content checksum behind the EndMark
};
// The u32 length hops over a frame's blocks (decompress.rs:205-235).  Block checksums are only collected here: the
// device hashes all blocks of a call in one launch and the delivery loop compares, in stream order.
void scan_blocks(const uint8_t* in, size_t in_len, const lzf_frame_info& fi, std::vector<Blk>& blocks, FrameScan& sc) {
    const size_t bmax = (size_t)fi.block_maxsize;
    const bool bsum = fi.flags & FL_BLOCKSUM, csum = fi.flags & FL_CSUM;
    size_t r = fi.header_len;
    for (;;) {
        if (in_len - r < 4) { sc.err = LZF_F_INPUT_ERROR; r = in_len; break; }
        uint32_t bl = rd32(in + r); r += 4;
        if (bl == 0) {                                                          // :206-215
            if (csum) { if (in_len - r < 4) { sc.err = LZF_F_INPUT_ERROR; r = in_len; break; } sc.want_content = rd32(in + r); r += 4; }
            sc.endmark = true; break;
        }
        const bool compressed = (bl & INCOMPRESSIBLE) == 0; bl &= ~INCOMPRESSIBLE;
        if (bl > (uint32_t)bmax) { sc.err = LZF_F_BLOCK_SIZE_OVERFLOW; break; }               // :220-222
        if (in_len - r < bl) { sc.err = LZF_F_INPUT_ERROR; r = in_len; break; }               // :226
        const uint8_t* data = in + r; r += bl;
        uint32_t c = 0;
        if (bsum) {                                                                           // :228-230
            if (in_len - r < 4) { sc.err = LZF_F_INPUT_ERROR; r = in_len; break; }
            c = rd32(in + r); r += 4;
        }
        blocks.push_back({data, bl, compressed, c, r});
    }
    sc.consumed = r;
}

}  // namespace

extern "C" {

uint32_t lzf_xxh32(const uint8_t* p, size_t len, uint32_t seed) { Xxh32 h(seed); h.update(p, len); return h.digest(); }

static_assert(sizeof(lzf_xxh32_state) >= sizeof(uint32_t) * 4 + 16 + 4 + 4 + 8, "state layout");
void lzf_xxh32_reset(lzf_xxh32_state* st, uint32_t seed) {
    Xxh32 h(seed);
    memcpy(st->v, h.v, sizeof st->v); st->fill = 0; st->seed = seed; st->total = 0;
}
void lzf_xxh32_update(lzf_xxh32_state* st, const uint8_t* p, size_t len) {
    Xxh32 h(st->seed);
    memcpy(h.v, st->v, sizeof h.v); memcpy(h.buf, st->buf, 16); h.fill = st->fill; h.total = st->total;
    h.update(p, len);
    memcpy(st->v, h.v, sizeof st->v); memcpy(st->buf, h.buf, 16); st->fill = h.fill; st->total = h.total;
}
uint32_t lzf_xxh32_digest(const lzf_xxh32_state* st) {
    Xxh32 h(st->seed);
    memcpy(h.v, st->v, sizeof h.v); memcpy(h.buf, st->buf, 16); h.fill = st->fill; h.total = st->total;
    return h.digest();
}

void lzf_settings_default(lzf_settings* s) {
    memset(s, 0, sizeof *s);
    s->independent_blocks = 1; s->block_checksums = 0; s->content_checksum = 1;
    s->block_size = 4u << 20;
}

size_t lzf_frame_compress_bound(const lzf_settings* s, size_t in_len) {
    const size_t bs = s->block_size ? (size_t)s->block_size : 1;
    return 19 + in_len + (in_len / bs + 1) * 8 + 8;
}

int lzf_frame_assemble(const lzf_settings* s, uint32_t n_blocks, const uint8_t* const* payload,
                       const uint32_t* comp_len, const uint32_t* raw_len, uint32_t content_xxh32,
                       uint8_t* out, size_t out_cap, size_t* out_len) {
    *out_len = 0;
    uint8_t bd;
    int rc = bd_new(s->block_size, &bd);
    if (rc != LZF_OK) return rc;
    size_t need = 19 + 8;
    for (uint32_t i = 0; i < n_blocks; ++i) need += 8 + (comp_len[i] == UINT32_MAX ? raw_len[i] : comp_len[i]);
    if (out_cap < need) return LZF_OUT_CAPACITY;
    size_t w = write_header(s, bd, out);
    for (uint32_t i = 0; i < n_blocks; ++i) {
        const bool stored = comp_len[i] == UINT32_MAX;
        const uint32_t n = stored ? raw_len[i] : comp_len[i];
        wr32(out + w, stored ? (n | INCOMPRESSIBLE) : n); w += 4;              // compress.rs:247,253
        memcpy(out + w, payload[i], n);                                        // :258
        if (s->block_checksums) { wr32(out + w + n, lzf_xxh32(out + w, n, 0)); ++g_host_block_hashes; }   // :259-263 (host payloads: host hash)
        w += n + (s->block_checksums ? 4 : 0);
    }
    wr32(out + w, 0); w += 4;                                                  // :277 EndMark
    if (s->content_checksum) { wr32(out + w, content_xxh32); w += 4; }         // :279-281
    *out_len = w;
    return LZF_OK;
}

// compress.rs:160-282 — one frame = a batch of one (lzf_frame_compress_many below)
int lzf_frame_compress(const lzf_settings* s, const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, size_t* out_len) {
    *out_len = 0;
    int st = LZF_OK;
    const int rc = lzf_frame_compress_many(s, 1, &in, &in_len, &out, &out_cap, out_len, &st);
    return rc != LZF_OK ? rc : st;
}

// decompress.rs:102-161; *consumed = bytes the reference's reader has read when it returns
static int read_header_ex(const uint8_t* in, size_t in_len, lzf_frame_info* info, size_t* consumed) {
    memset(info, 0, sizeof *info);
    size_t r = 0;
    int rc = LZF_OK;
#define NEED(n) do { if (in_len - r < (size_t)(n)) { r = in_len; rc = LZF_F_INPUT_ERROR; goto done; } } while (0)
#define FAIL(c) do { rc = (c); goto done; } while (0)
    {
    NEED(4); r = 4; if (rd32(in) != LZF_MAGIC) FAIL(LZF_F_WRONG_MAGIC);       // :103-106
    NEED(1); const uint8_t flg = in[r++];
    if ((flg >> 6) != 1) FAIL(LZF_F_UNSUPPORTED_VERSION);                     // header.rs:33-36
    if (flg & 0x02) FAIL(LZF_F_RESERVED_FLAG_BITS);                           // header.rs:37-39
    NEED(1); const uint8_t bd = in[r++];
    if (bd & 0x8F) FAIL(LZF_F_RESERVED_BD_BITS);                              // header.rs:66-68
    info->flags = flg; info->bd = bd;
    if (flg & FL_CSIZE) { NEED(8); info->has_content_size = 1; info->content_size = (uint64_t)rd32(in + r) | ((uint64_t)rd32(in + r + 4) << 32); r += 8; }
    if (flg & FL_DICTID) { NEED(4); info->has_dictionary_id = 1; info->dictionary_id = rd32(in + r); r += 4; }
    NEED(1); const uint8_t hc = in[r++];
    if (hc != (uint8_t)(lzf_xxh32(in + 4, r - 5, 0) >> 8)) FAIL(LZF_F_HEADER_CHECKSUM_FAIL);   // :132-136
    const unsigned size = (bd >> 4) & 7;
    if (size < 4) FAIL(LZF_F_UNIMPLEMENTED_BLOCKSIZE);                        // :153, header.rs:73-80
    info->block_maxsize = 1ull << (size * 2 + 8);
    info->header_len = (uint16_t)r;
    }
done:
    if (consumed) *consumed = r;
    return rc;
#undef NEED
#undef FAIL
}
int lzf_frame_read_header(const uint8_t* in, size_t in_len, lzf_frame_info* info) { return read_header_ex(in, in_len, info, nullptr); }

// decompress.rs:198-288 — one frame = a batch of one (lzf_frame_decompress_many below)
int lzf_frame_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len,
                         uint8_t* out, size_t out_cap, size_t* out_len, size_t* consumed) {
    *out_len = 0; if (consumed) *consumed = 0;
    int st = LZF_OK;
    const int rc = lzf_frame_decompress_many(1, &in, &in_len, dict, dict_len, &out, &out_cap, out_len, consumed, &st);
    return rc != LZF_OK ? rc : st;
}

// ---- the block-by-block reader: LZ4FrameReader (decompress.rs:79-282) ------------------------------------------------
}  // extern "C"
struct lzf_frame_reader {
    const uint8_t* in; size_t in_len, pos;
    lzf_frame_info fi;
    bool finished = false, has_hasher = false, linked = false;
    Xxh32 hasher{0};                    // content_hasher :89,:140-142
    std::vector<uint8_t> window;        // carryover_window :90,:144-148
};
extern "C" {

int lzf_frame_reader_new(const uint8_t* in, size_t in_len, lzf_frame_reader** r) {
    if (!r) return LZF_E_INVALID;
    *r = nullptr;
    lzf_frame_info fi;
    const int rc = lzf_frame_read_header(in, in_len, &fi);
    if (rc != LZF_OK) return rc;
    lzf_frame_reader* rd = new lzf_frame_reader;
    rd->in = in; rd->in_len = in_len; rd->pos = fi.header_len; rd->fi = fi;
    rd->has_hasher = (fi.flags & FL_CSUM) != 0; rd->linked = !(fi.flags & FL_INDEP);
    *r = rd;
    return LZF_OK;
}
void lzf_frame_reader_free(lzf_frame_reader* r) { delete r; }
void lzf_frame_reader_info(const lzf_frame_reader* r, lzf_frame_info* info) { *info = r->fi; }
int lzf_frame_reader_finished(const lzf_frame_reader* r) { return r->finished ? 1 : 0; }
size_t lzf_frame_reader_consumed(const lzf_frame_reader* r) { return r->pos; }

int lzf_frame_reader_decode_block(lzf_frame_reader* r, const uint8_t* dict, size_t dict_len, uint8_t* out, size_t out_cap, size_t* out_len) {
    if (!r || !out_len || (!out && out_cap)) return LZF_E_INVALID;
    *out_len = 0;
    if (!dict) dict_len = 0;
    if (r->finished) return LZF_OK;                                             // :202
#define NEED(n) do { if (r->in_len - r->pos < (size_t)(n)) { r->pos = r->in_len; return LZF_F_INPUT_ERROR; } } while (0)
    NEED(4); uint32_t bl = rd32(r->in + r->pos); r->pos += 4;                   // :205
    if (bl == 0) {                                                              // :206-215
        if (r->has_hasher) {
            r->has_hasher = false;                                              // content_hasher.take()
            NEED(4); const uint32_t c = rd32(r->in + r->pos); r->pos += 4;
            if (c != r->hasher.digest()) return LZF_F_FRAME_CHECKSUM_FAIL;
        }
        r->finished = true;
        return LZF_OK;
    }
    const bool compressed = (bl & INCOMPRESSIBLE) == 0; bl &= ~INCOMPRESSIBLE;  // :217-218
    const size_t bmax = (size_t)r->fi.block_maxsize;
    if (bl > (uint32_t)bmax) return LZF_F_BLOCK_SIZE_OVERFLOW;                  // :220-222
    NEED(bl); const uint8_t* data = r->in + r->pos; r->pos += bl;               // :224-226
    uint32_t want = 0; const bool bsum = (r->fi.flags & FL_BLOCKSUM) != 0;
    if (bsum) { NEED(4); want = rd32(r->in + r->pos); r->pos += 4; }            // :229
#undef NEED
    // the prefix (:238-245)
    const uint8_t* prefix = dict; size_t prefix_len = dict_len;
    if (r->linked) {
        if (r->window.empty() && dict_len) r->window.assign(dict, dict + dict_len);
        prefix = r->window.data(); prefix_len = r->window.size();
    }
    // the block goes to the device once: checksum (:228-235) and decode (:247-251) both read it there
    lzf_host::Staging& sg = lzf_host::Staging::get();
    std::lock_guard<std::mutex> guard(sg.lock());
    hipStream_t cs = sg.stream(0);
    const size_t cap = bmax + bl;                                               // limit + what the literals may overshoot
    uint8_t* d_in = static_cast<uint8_t*>(sg.device(0, (size_t)bl + 8));
    uint8_t* d_out = static_cast<uint8_t*>(sg.device(1, cap));
    uint8_t* d_pre = static_cast<uint8_t*>(sg.device(10, prefix_len));
    uint8_t* d_meta = static_cast<uint8_t*>(sg.device(5, 1024));
    if (!cs || !d_in || !d_out || !d_pre || !d_meta) return fail_hip();
#define HIPR(e) do { if ((e) != hipSuccess) { (void)hipDeviceSynchronize(); return LZF_E_HIP; } } while (0)
    HIPR(hipMemcpyAsync(d_in, data, bl, hipMemcpyHostToDevice, cs));
    if (bsum) {
        struct { const uint8_t* p; uint64_t n; uint32_t h; } m = {d_in, bl, 0};
        HIPR(hipMemcpyAsync(d_meta, &m, sizeof m, hipMemcpyHostToDevice, cs));
        int rc = lzf_xxh32_batch(reinterpret_cast<const uint8_t* const*>(d_meta), reinterpret_cast<const uint64_t*>(d_meta + 8), reinterpret_cast<uint32_t*>(d_meta + 16), 1, cs);
        if (rc != LZF_OK) return rc;
        uint32_t got = 0;
        HIPR(hipMemcpyAsync(&got, d_meta + 16, 4, hipMemcpyDeviceToHost, cs));
        HIPR(hipStreamSynchronize(cs));
        ++g_reader_device_hashes;
        if (got != want) return LZF_F_BLOCK_CHECKSUM_FAIL;
    }
    size_t n = 0;
    if (compressed) {                                                           // :247-248
        if (prefix_len) HIPR(hipMemcpyAsync(d_pre, prefix, prefix_len, hipMemcpyHostToDevice, cs));
        lzf_decompress_job j; memset(&j, 0, sizeof j);
        j.input = d_in; j.input_len = bl; j.prefix = d_pre; j.prefix_len = prefix_len; j.out = d_out; j.out_cap = cap; j.output_limit = bmax;
        lzf_job_result res; memset(&res, 0, sizeof res);
        HIPR(hipMemcpyAsync(d_meta + 64, &j, sizeof j, hipMemcpyHostToDevice, cs));
        int rc = lzf_decompress_batch_sized(reinterpret_cast<lzf_decompress_job*>(d_meta + 64), reinterpret_cast<lzf_job_result*>(d_meta + 256), 1, bl, cs);
        if (rc != LZF_OK) return rc;
        HIPR(hipMemcpyAsync(&res, d_meta + 256, sizeof res, hipMemcpyDeviceToHost, cs));
        HIPR(hipStreamSynchronize(cs));
        if (res.status != LZF_OK) return res.status;                            // CodecError
        n = (size_t)res.out_len;
        if (n > out_cap) return LZF_OUT_CAPACITY;
        if (n) { HIPR(hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, cs)); HIPR(hipStreamSynchronize(cs)); }
    } else {                                                                    // :249-251
        n = bl;
        if (n > out_cap) return LZF_OUT_CAPACITY;
        HIPR(hipStreamSynchronize(cs));
        memcpy(out, data, n);
    }
#undef HIPR
    *out_len = n;
    if (r->linked) {                                                            // :253-269
        std::vector<uint8_t>& w = r->window;
        if (n < LZF_WINDOW_SIZE) {
            const size_t avail = w.size() + n;
            if (avail >= LZF_WINDOW_SIZE) w.erase(w.begin(), w.begin() + (avail - LZF_WINDOW_SIZE));
            w.insert(w.end(), out, out + n);
        } else w.assign(out + n - LZF_WINDOW_SIZE, out + n);
    }
    if (n > bmax) return LZF_F_BLOCK_SIZE_OVERFLOW;                             // :272-274
    if (r->has_hasher) r->hasher.update(out, n);                                // :276-278
    return LZF_OK;
}

// =====================================================================================================================
// Many frames per call.  One frame of a few large blocks leaves the chip almost empty (one wavefront per block); the
// batch entry points want thousands of blocks.  These drivers put every block of every frame into the same launches
// and keep inputs, tables and outputs on the device from the first block to the last: independent-block frames are
// one launch; linked-block frames advance together, block k of every stream in launch k, with no host round trip in
// between (the jobs of all steps are known up front on the compress side — the window is input data and the table
// lives on the device; on the decompress side lzf_chain_decompress_step patches each stream's length on the device).
// Same bytes and the same statuses as calling lzf_frame_compress / lzf_frame_decompress once per frame.
//
// Data movement (host_staging.h): the caller's buffers go through one pinned slab in 4 MiB pieces, worker threads doing the
// memcpy while the calling thread issues one asynchronous DMA per finished piece; results come back the same way from a
// buffer the device has packed (lzf_copy_ranges), so only the bytes that are part of the result cross PCIe.  Block
// checksums (compress.rs:259-263, decompress.rs:228-235) are computed by lzf_xxh32_batch on the device for all blocks at
// once; content checksums (compress.rs:233-235,279-281; decompress.rs:207-211,276-278) likewise for frames up to
// kDeviceHashMax bytes — XXH32 is one serial chain per buffer, ~1.3 GB/s per chain on the GPU but thousands of chains at
// once, against ~6 GB/s on one host core — and on worker threads (one frame per thread, overlapping the kernels) beyond.
// =====================================================================================================================
}  // extern "C"
namespace {
using lzf_host::Seg;
using lzf_host::Staging;
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
// (on failure: nothing asynchronous may still be reading the host arrays of the frame that returns)
#define HIPOK(e) do { if ((e) != hipSuccess) { (void)hipDeviceSynchronize(); return LZF_E_HIP; } } while (0)
#define RCOK(e) do { const int rc__ = (e); if (rc__ != LZF_OK) { (void)hipDeviceSynchronize(); return rc__; } } while (0)
constexpr size_t kDeviceHashMax = 32u << 20;
#ifdef LZF_ANALYSIS      // LZF_FRAME_TRACE=1: wall-clock milliseconds between the marks of a *_many call, on stderr
struct Trace {
    bool on; std::chrono::steady_clock::time_point t0;
    Trace() : on(getenv("LZF_FRAME_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) { if (!on) return; const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[frame] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count()); t0 = t; }
};
#define TRACE_BEGIN() Trace trace__
#define TRACE(what) trace__.mark(what)
#else
#define TRACE_BEGIN() do {} while (0)
#define TRACE(what) do {} while (0)
#endif

// device scratch slots of the frame drivers (Staging::device)
enum { S_IN = 0, S_OUT, S_PACK, S_JOBS, S_RES, S_LISTS, S_TABS, S_TABPTR, S_ADDS, S_TMPL, S_DICT, S_STEPS, S_STATE, S_HASH, S_LISTS_G /* .. + 7: per group */, S_IN_G = S_LISTS_G + 8, S_OUT_G = S_IN_G + 8, S_COUNT = S_OUT_G + 8 };
static_assert(S_COUNT <= Staging::kSlots, "device scratch slots");

// counters of lzf_frame_get_stats (calls on different devices run concurrently)
struct { std::atomic<uint64_t> calls{0}, device_block_hashes{0}, device_content_hashes{0}, host_content_hashes{0}; } g_stats;

// n small host arrays -> one device slot, one copy: ptr(i) is array i on the device
struct Lists {
    std::vector<uint8_t> h; std::vector<size_t> off; uint8_t* d = nullptr;
    size_t add(const void* p, size_t bytes) { const size_t o = h.size(); h.resize(up256(o + bytes)); if (bytes && p) memcpy(h.data() + o, p, bytes); off.push_back(o); return off.size() - 1; }
    int upload(Staging& sg, int slot, hipStream_t st) {
        d = static_cast<uint8_t*>(sg.device(slot, h.size()));
        if (!d) return fail_hip();
        if (!h.empty()) HIPOK(hipMemcpyAsync(d, h.data(), h.size(), hipMemcpyHostToDevice, st));
        return LZF_OK;
    }
    template <class T> T* ptr(size_t i) const { return reinterpret_cast<T*>(d + off[i]); }
};

// XXH32 of n host buffers on the worker threads (frames too long for one device chain each)
void host_hashes(Staging& sg, const std::vector<const uint8_t*>& p, const std::vector<size_t>& n, std::vector<uint32_t>& out) {
    out.resize(p.size());
    sg.parallel_for(p.size(), [&](size_t i) { out[i] = lzf_xxh32(p[i], n[i], 0); });
    g_stats.host_content_hashes += p.size();
}
}  // namespace
extern "C" {

void lzf_frame_get_stats(lzf_frame_stats* st) {
    Staging& sg = Staging::get();
    std::lock_guard<std::mutex> g(sg.lock());
    memset(st, 0, sizeof *st);
    st->calls = g_stats.calls; st->device_content_hashes = g_stats.device_content_hashes; st->host_content_hashes = g_stats.host_content_hashes;
    st->host_block_hashes = g_host_block_hashes;
    st->device_block_hashes = g_stats.device_block_hashes + g_reader_device_hashes;
    st->h2d_copies = sg.counters.h2d_copies; st->d2h_copies = sg.counters.d2h_copies;
    st->h2d_bytes = sg.counters.h2d_bytes; st->d2h_bytes = sg.counters.d2h_bytes;
    st->pinned_bytes = sg.pinned_capacity();
}
void lzf_frame_release_scratch(void) {
    Staging& sg = Staging::get();
    std::lock_guard<std::mutex> g(sg.lock());
    sg.release();
}
void lzf_frame_set_host_threads(uint32_t n) {
    Staging& sg = Staging::get();
    std::lock_guard<std::mutex> g(sg.lock());
    sg.set_threads(n);
}

void lzf_frame_set_pinned_limit(size_t bytes) {
    Staging& sg = Staging::get();
    std::lock_guard<std::mutex> g(sg.lock());
    sg.set_pinned_limit(bytes);
}

// One pass of lzf_frame_compress_many: the frames given travel together (one pinned slab, one set of device buffers).
static int compress_many_pass(const lzf_settings* s, uint32_t n_frames, const uint8_t* const* in, const size_t* in_len,
                              uint8_t* const* out, const size_t* out_cap, size_t* out_len, int* status);

// Device memory of a call is bounded: the frames go through in passes of at most a third of the memory budget
// (lzf_frame_set_memory_budget; by default half of the free HBM) of input bytes — input, output slots and the packed result
// of a pass are on the device together; a single frame larger than that is a pass of its own.  Pinned host memory is bounded
// by the staging itself (host_staging.cpp: the slab stops growing at 2 GiB and is recycled as a ring within a pass), so a pass
// is as large as the device allows: until round 6 a pass was also capped at 2 GiB of pinned memory, and a 16 GiB call of 4 MiB
// blocks went through as eight passes of 512 blocks — each of them in the single-block latency class of the compress kernel.
constexpr size_t kSmallCall = (size_t)1 << 30;      // calls up to this size are one pass without asking the device

int lzf_frame_compress_many(const lzf_settings* s, uint32_t n_frames, const uint8_t* const* in, const size_t* in_len,
                            uint8_t* const* out, const size_t* out_cap, size_t* out_len, int* status) {
    if (!s || (n_frames && (!in || !in_len || !out || !out_cap || !out_len || !status))) return LZF_E_INVALID;
    size_t total = 0;
    for (uint32_t f = 0; f < n_frames; ++f) total += in_len[f];
    size_t limit = kSmallCall;
    if (total > limit || g_budget) {                                            // (small calls skip the device query)
        size_t free_b = 0, total_b = 0;
        if (g_budget) limit = g_budget / 3;                                     // input + output slots + packed result per pass
        else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) limit = free_b / 2 / 3;
        else (void)hipGetLastError();
    }
    if (total <= limit) return compress_many_pass(s, n_frames, in, in_len, out, out_cap, out_len, status);
    for (uint32_t f0 = 0; f0 < n_frames;) {
        size_t sum = 0; uint32_t f1 = f0;
        while (f1 < n_frames && (f1 == f0 || sum + in_len[f1] <= limit)) { sum += in_len[f1]; ++f1; }
        const int rc = compress_many_pass(s, f1 - f0, in + f0, in_len + f0, out + f0, out_cap + f0, out_len + f0, status + f0);
        if (rc != LZF_OK) return rc;
        f0 = f1;
    }
    return LZF_OK;
}

static int compress_many_pass(const lzf_settings* s, uint32_t n_frames, const uint8_t* const* in, const size_t* in_len,
                              uint8_t* const* out, const size_t* out_cap, size_t* out_len, int* status) {
    uint8_t bd;
    const int bd_rc = bd_new(s->block_size, &bd);                              // compress.rs:183
    for (uint32_t f = 0; f < n_frames; ++f) { out_len[f] = 0; status[f] = bd_rc != LZF_OK ? bd_rc : out_cap[f] < lzf_frame_compress_bound(s, in_len[f]) ? LZF_OUT_CAPACITY : LZF_OK; }
    if (bd_rc != LZF_OK || n_frames == 0) return LZF_OK;
    const size_t bs = (size_t)s->block_size;
    const uint8_t* dict = s->dictionary;
    const size_t dict_len = dict ? (size_t)s->dictionary_len : 0;
    const bool indep = s->independent_blocks != 0;
    const bool per_block_prefix = indep && dict_len > 0;                       // :218,:268 in_buffer = dict ++ block, for every block
    const bool bsum = s->block_checksums != 0, csum = s->content_checksum != 0;

    // ---- layout: input slab, output slab (cap n per block, :242), job list ordered by step
    struct Fr { size_t nb, in_off, job0, data_off, out0; uint32_t grp; };      // data_off: the frame's own bytes in the slab (not per_block_prefix); out0: its first output slot
    std::vector<Fr> fr(n_frames);
    size_t in_total = 0, out_total = 0, n_jobs = 0, max_nb = 0, pack_bound = 0;
    for (uint32_t f = 0; f < n_frames; ++f) {
        fr[f].nb = status[f] == LZF_OK ? (in_len[f] + bs - 1) / bs : 0;
        fr[f].in_off = in_total; fr[f].data_off = in_total + (per_block_prefix ? 0 : dict_len);
        fr[f].job0 = n_jobs; fr[f].out0 = pack_bound; fr[f].grp = 0;           // (job0: independent mode, where jobs are in frame order)
        if (fr[f].nb) in_total = up256(in_total + (per_block_prefix ? fr[f].nb * dict_len : dict_len) + in_len[f]);
        if (fr[f].nb) pack_bound += in_len[f];                                  // every block's output slot is as long as the block (:242)
        n_jobs += fr[f].nb; if (fr[f].nb > max_nb) max_nb = fr[f].nb;
    }
    if (n_jobs == 0) {                                                          // only empty inputs: header + EndMark each
        for (uint32_t f = 0; f < n_frames; ++f) if (status[f] == LZF_OK) {
            uint32_t content = 0; if (csum) content = lzf_xxh32(in[f], 0, 0);
            status[f] = lzf_frame_assemble(s, 0, nullptr, nullptr, nullptr, content, out[f], out_cap[f], &out_len[f]);
        }
        return LZF_OK;
    }
    if (n_jobs > 0x7FFFFFFFull) return LZF_E_INVALID;
    Staging& sg = Staging::get();
    std::lock_guard<std::mutex> guard(sg.lock());
    TRACE_BEGIN();
    hipStream_t cs = sg.stream(0), hs = sg.stream(3);
    if (!cs || !hs) return fail_hip();
    if (!sg.pinned(in_total > pack_bound ? in_total : pack_bound)) return fail_hip();
    TRACE("c: pinned slab");
    std::vector<lzf_compress_job> jobs(n_jobs);
    std::vector<uint32_t> job_frame(n_jobs), job_block(n_jobs);
    std::vector<size_t> job_out_off(n_jobs);
    std::vector<size_t> step_off;                                               // linked: jobs of step k are [step_off[k], step_off[k+1])
    // ---- sub-groups of whole frames.  A call with enough independent blocks to fill the chip several times over travels in up to
    //      kPipe groups, one behind the other on the same stream: group g compresses while g + 1 is still on its way in and
    //      g - 1 is on its way out.  (Linked streams, and calls too small to keep the chip busy per group, are one group.)
    //      Every group has its own input and output allocation: the runtime orders a copy behind the last launch that touches
    //      the same allocation, whatever the stream, so with one buffer for all no copy would overlap a kernel.
    constexpr uint32_t kPipe = 8, kFill = 4608;                                 // kFill: one-wave jobs the chip holds at once (18 per CU)
    // (groups of at most kFill jobs where kPipe groups allow it: such a launch needs no cost-ordered launch list, whose scratch
    //  allocation would wait for the group before it)
    const uint32_t n_groups = indep && n_jobs >= 2u * kFill ? ((n_jobs + kFill - 1) / kFill < kPipe ? (uint32_t)((n_jobs + kFill - 1) / kFill) : kPipe) : 1u;
    std::vector<uint32_t> gend(n_groups, n_frames);                             // group g = frames [gend[g-1], gend[g])
    for (uint32_t g = 0, f = 0; g + 1 < n_groups; ++g) {
        while (f < n_frames && (fr[f].nb == 0 || fr[f].job0 + fr[f].nb < n_jobs * (size_t)(g + 1) / n_groups)) ++f;
        gend[g] = f < n_frames ? ++f : n_frames;
    }
    // the layout is one address space for inputs [0, in_total) and one for output slots [0, pack_bound); group g's part of each is
    // backed by its own allocation, addressed through a base biased by where the part starts
    std::vector<uintptr_t> din_b(n_groups), dout_b(n_groups);
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t fa = g ? gend[g - 1] : 0, fb = gend[g];
        const size_t ia = fa < n_frames ? fr[fa].in_off : in_total, ib = fb < n_frames ? fr[fb].in_off : in_total;
        const size_t oa = fa < n_frames ? fr[fa].out0 : pack_bound, ob = fb < n_frames ? fr[fb].out0 : pack_bound;
        void* const di = sg.device(S_IN_G + (int)g, ib - ia);
        void* const dob = sg.device(S_OUT_G + (int)g, ob - oa);
        if (!di || !dob) return fail_hip();
        din_b[g] = reinterpret_cast<uintptr_t>(di) - ia; dout_b[g] = reinterpret_cast<uintptr_t>(dob) - oa;
        for (uint32_t f = fa; f < fb; ++f) fr[f].grp = g;
    }
    auto din_at = [&](uint32_t f, size_t off) { return reinterpret_cast<uint8_t*>(din_b[fr[f].grp] + off); };   // off: offset in the input address space
    std::vector<Seg> up;                                                        // host -> slab, in frame order
    std::vector<size_t> up_first(n_frames, 0);                                  // independent mode: frame f's pieces start at up[up_first[f]]
    auto seg = [&](size_t off, const uint8_t* p, size_t n) { if (n) up.push_back({off, const_cast<uint8_t*>(p), n}); };

    lzf_u32_table tmpl;
    if (dict_len >= 8) { RCOK(seeded_template(dict, dict_len, &tmpl)); } else memset(&tmpl, 0, sizeof tmpl);
    std::vector<void*> h_tabptr(n_jobs, nullptr);
    std::vector<uint64_t> h_adds(n_jobs, 0);
    uint32_t n_linked = 0;
    std::vector<uint32_t> lf_index(n_frames, 0);
    lzf_u32_table* d_tabs = nullptr; void* d_tmpl = nullptr;
    if (!indep) {
        for (uint32_t f = 0; f < n_frames; ++f) if (fr[f].nb) lf_index[f] = n_linked++;
        d_tabs = static_cast<lzf_u32_table*>(sg.device(S_TABS, sizeof(lzf_u32_table) * (size_t)(n_linked ? n_linked : 1)));
        if (!d_tabs) return fail_hip();
    } else if (dict_len >= 8) {
        d_tmpl = sg.device(S_TMPL, sizeof tmpl);
        if (!d_tmpl) return fail_hip();
        HIPOK(hipMemcpyAsync(d_tmpl, &tmpl, sizeof tmpl, hipMemcpyHostToDevice, cs));
    }

    size_t jn = 0;
    if (indep) {
        for (uint32_t f = 0; f < n_frames; ++f) {
            fr[f].job0 = jn; up_first[f] = up.size();
            size_t w = fr[f].in_off;
            if (fr[f].nb && !per_block_prefix) seg(w, in[f], in_len[f]);
            for (size_t i = 0; i < fr[f].nb; ++i, ++jn) {
                const size_t off = i * bs, n = in_len[f] - off < bs ? in_len[f] - off : bs;
                lzf_compress_job& j = jobs[jn];
                memset(&j, 0, sizeof j);
                if (per_block_prefix) {
                    seg(w, dict, dict_len); seg(w + dict_len, in[f] + off, n);
                    j.input = din_at(f, w); j.input_len = dict_len + n; j.cursor = dict_len; w += dict_len + n;
                } else { j.input = din_at(f, fr[f].in_off + off); j.input_len = n; j.cursor = 0; }
                j.out_cap = n; j.table_kind = LZF_TABLE_U32;                   // :242, :202
                if (dict_len >= 8) { j.table = d_tmpl; j.flags = LZF_CJOB_TABLE_READONLY; }       // :220,:270 template.clone()
                job_frame[jn] = f; job_block[jn] = (uint32_t)i; job_out_off[jn] = out_total; out_total += n;
            }
        }
        step_off = {0, n_jobs};
    } else {
        // linked blocks (:271-275): in_buffer = the last <= 64 KiB of (dict ++ data so far) ++ block, a pointer into the
        // stream's slab; the table's offset grows by what the window forgets
        struct Ls { size_t lo, len; };                                          // in_buffer = slab[lo, lo + len)
        std::vector<Ls> ls(n_frames, Ls{0, dict_len});
        for (uint32_t f = 0; f < n_frames; ++f) if (fr[f].nb) { seg(fr[f].in_off, dict, dict_len); seg(fr[f].in_off + dict_len, in[f], in_len[f]); }
        std::vector<uint64_t> pending_add(n_frames, 0);
        for (size_t k = 0; k < max_nb; ++k) {
            step_off.push_back(jn);
            for (uint32_t f = 0; f < n_frames; ++f) {
                if (k >= fr[f].nb) continue;
                const size_t off = k * bs, n = in_len[f] - off < bs ? in_len[f] - off : bs;
                lzf_compress_job& j = jobs[jn];
                memset(&j, 0, sizeof j);
                j.input = din_at(f, fr[f].in_off + ls[f].lo); j.input_len = ls[f].len + n; j.cursor = ls[f].len;   // :222,:243
                j.out_cap = n; j.table_kind = LZF_TABLE_U32;
                j.table = d_tabs + lf_index[f];
                h_tabptr[jn] = j.table; h_adds[jn] = pending_add[f];           // applied before this step
                job_frame[jn] = f; job_block[jn] = (uint32_t)k; job_out_off[jn] = out_total; out_total += n;
                ++jn;
                ls[f].len += n;
                pending_add[f] = 0;
                if (ls[f].len > LZF_WINDOW_SIZE) { const size_t forget = ls[f].len - LZF_WINDOW_SIZE; pending_add[f] = forget; ls[f].lo += forget; ls[f].len = LZF_WINDOW_SIZE; }
            }
        }
        step_off.push_back(jn);
    }
    lzf_compress_job* const d_jobs = static_cast<lzf_compress_job*>(sg.device(S_JOBS, sizeof(lzf_compress_job) * n_jobs));
    lzf_job_result* const d_res = static_cast<lzf_job_result*>(sg.device(S_RES, sizeof(lzf_job_result) * n_jobs));
    if (!d_jobs || !d_res) return fail_hip();
    for (size_t q = 0; q < n_jobs; ++q) jobs[q].out = reinterpret_cast<uint8_t*>(dout_b[fr[job_frame[q]].grp] + job_out_off[q]);
    // results come back through the pinned mailbox: [block results | content hashes | block checksums]
    const size_t mb_res = 0, mb_chash = up256(sizeof(lzf_job_result) * n_jobs), mb_sums = mb_chash + up256(sizeof(uint32_t) * n_frames);
    uint8_t* const mbox = sg.mailbox(mb_sums + sizeof(uint32_t) * n_jobs);
    if (!mbox) return fail_hip();
    const lzf_job_result* const res = reinterpret_cast<const lzf_job_result*>(mbox + mb_res);
    TRACE("c: layout + scratch");
    // ---- in: small arrays first, then group after group: its bytes (pieces, asynchronous), its launches, its results
    HIPOK(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(lzf_compress_job) * n_jobs, hipMemcpyHostToDevice, cs));
    void** d_tabptr = nullptr; uint64_t* d_adds = nullptr;
    std::vector<lzf_u32_table> h_tabs;
    if (!indep) {
        h_tabs.assign(n_linked, tmpl);                                          // :213-214 table = template.clone()
        HIPOK(hipMemcpyAsync(d_tabs, h_tabs.data(), sizeof(lzf_u32_table) * n_linked, hipMemcpyHostToDevice, cs));
        d_tabptr = static_cast<void**>(sg.device(S_TABPTR, sizeof(void*) * n_jobs));
        d_adds = static_cast<uint64_t*>(sg.device(S_ADDS, sizeof(uint64_t) * n_jobs));
        if (!d_tabptr || !d_adds) return fail_hip();
        HIPOK(hipMemcpyAsync(d_tabptr, h_tabptr.data(), sizeof(void*) * n_jobs, hipMemcpyHostToDevice, cs));
        HIPOK(hipMemcpyAsync(d_adds, h_adds.data(), sizeof(uint64_t) * n_jobs, hipMemcpyHostToDevice, cs));
    }
    struct Events {                                                             // one "group g is compressed" event each
        std::vector<hipEvent_t> e;
        ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } done;
    done.e.assign(n_groups, nullptr);
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t fa = g ? gend[g - 1] : 0, fb = gend[g];
        const size_t ua = fa < n_frames ? up_first[fa] : up.size(), ub = fb < n_frames ? up_first[fb] : up.size();
        HIPOK(sg.upload(std::vector<Seg>(up.begin() + ua, up.begin() + ub), in_total, reinterpret_cast<uint8_t*>(din_b[g])));
        TRACE("c:   group upload");
        HIPOK(sg.join_copies(cs));
        TRACE("c:     join");
        if (indep) {
            size_t a = n_jobs, e = 0;
            for (uint32_t f = fa; f < fb; ++f) if (fr[f].nb) { if (fr[f].job0 < a) a = fr[f].job0; if (fr[f].job0 + fr[f].nb > e) e = fr[f].job0 + fr[f].nb; }
            if (e > a) {
                RCOK(lzf_compress_batch(d_jobs + a, d_res + a, (uint32_t)(e - a), LZF_KINDS_U32 | LZF_KINDS_U32_FRESH_ONLY, cs));
                TRACE("c:     batch call");
                HIPOK(hipMemcpyAsync(mbox + mb_res + sizeof(lzf_job_result) * a, d_res + a, sizeof(lzf_job_result) * (e - a), hipMemcpyDeviceToHost, cs));
                TRACE("c:     result copy");
            }
        } else {
            for (size_t k = 0; k + 1 < step_off.size(); ++k) {                 // no host round trip between the steps
                const size_t a = step_off[k], cnt = step_off[k + 1] - a;
                if (!cnt) continue;
                if (k > 0) RCOK(lzf_table_offset_batch(d_tabptr + a, d_adds + a, (uint32_t)cnt, LZF_TABLE_U32, cs));
                RCOK(lzf_compress_batch(d_jobs + a, d_res + a, (uint32_t)cnt, LZF_KINDS_U32, cs));
            }
            HIPOK(hipMemcpyAsync(mbox + mb_res, d_res, sizeof(lzf_job_result) * n_jobs, hipMemcpyDeviceToHost, cs));
        }
        HIPOK(hipEventCreateWithFlags(&done.e[g], hipEventDisableTiming));
        HIPOK(hipEventRecord(done.e[g], cs));
        TRACE("c:   group launch");
    }
    TRACE("c: uploads + launches issued");
    // ---- content checksums (:233-235): one device chain per frame on the checksum stream while the blocks compress;
    //      long frames (and dict ++ block layouts, where the frame is not contiguous in the slab) on the workers
    std::vector<uint32_t> content(n_frames, 0);
    std::vector<uint32_t> dev_hash_frames, host_hash_frames;
    if (csum) for (uint32_t f = 0; f < n_frames; ++f) if (status[f] == LZF_OK) {
        if (fr[f].nb == 0) content[f] = lzf_xxh32(in[f], 0, 0);
        else if (!per_block_prefix && in_len[f] <= kDeviceHashMax) dev_hash_frames.push_back(f);
        else host_hash_frames.push_back(f);
    }
    Lists hl; uint32_t* d_chash = nullptr;
    const uint32_t* const chash = reinterpret_cast<const uint32_t*>(mbox + mb_chash);
    if (!dev_hash_frames.empty()) {
        std::vector<const uint8_t*> p; std::vector<uint64_t> n;
        for (uint32_t f : dev_hash_frames) { p.push_back(din_at(f, fr[f].data_off)); n.push_back(in_len[f]); }
        const size_t ip = hl.add(p.data(), p.size() * sizeof p[0]), il = hl.add(n.data(), n.size() * sizeof n[0]), io = hl.add(nullptr, p.size() * sizeof(uint32_t));
        HIPOK(sg.join_copies(hs));
        RCOK(hl.upload(sg, S_HASH, hs));
        d_chash = hl.ptr<uint32_t>(io);
        RCOK(lzf_xxh32_batch(hl.ptr<const uint8_t*>(ip), hl.ptr<uint64_t>(il), d_chash, (uint32_t)p.size(), hs));
        HIPOK(hipMemcpyAsync(mbox + mb_chash, d_chash, p.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
        g_stats.device_content_hashes += p.size();
    }
    if (!host_hash_frames.empty()) {
        std::vector<const uint8_t*> p; std::vector<size_t> n; std::vector<uint32_t> h;
        for (uint32_t f : host_hash_frames) { p.push_back(in[f]); n.push_back(in_len[f]); }
        host_hashes(sg, p, n, h);
        for (size_t i = 0; i < h.size(); ++i) content[host_hash_frames[i]] = h[i];
    }
    TRACE("c: content hashes issued / host hashes");
    // ---- out, group after group (:244-263, :277-281): header and length words straight into the caller's buffer, every
    //      compressed payload from its output slot through the slab to its place in the frame, stored blocks (:250-255)
    //      from the caller's own input, block checksums over the device copies of both
    std::vector<std::vector<size_t>> frame_jobs(n_frames);
    for (size_t q = 0; q < n_jobs; ++q) { auto& v = frame_jobs[job_frame[q]]; if (v.size() <= job_block[q]) v.resize(job_block[q] + 1); v[job_block[q]] = q; }
    struct Patch { uint8_t* at; size_t hash_index; };
    std::vector<Patch> sum_at;                                                  // where block checksum i goes
    struct HostCopy { uint8_t* dst; const uint8_t* src; size_t n; };
    std::vector<HostCopy> stored_copies;
    std::vector<uint8_t*> content_at(n_frames, nullptr);
    std::vector<Lists> sum_lists(n_groups);
    size_t n_hashed = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
        HIPOK(hipEventSynchronize(done.e[g]));                                  // the group's block results are in the mailbox
        if (g == 0) TRACE("c: first group compressed");
        std::vector<Seg> down;
        std::vector<const uint8_t*> h_ptr; std::vector<uint64_t> h_len;
        const size_t hash0 = n_hashed;
        for (uint32_t f = g ? gend[g - 1] : 0; f < gend[g]; ++f) {
            if (status[f] != LZF_OK) continue;
            if (fr[f].nb == 0) { status[f] = lzf_frame_assemble(s, 0, nullptr, nullptr, nullptr, content[f], out[f], out_cap[f], &out_len[f]); continue; }   // an empty input: header + EndMark
            const size_t nb = fr[f].nb;
            int st = LZF_OK;
            for (size_t i = 0; i < nb && st == LZF_OK; ++i) { const int bst = res[frame_jobs[f][i]].status; if (bst != LZF_OK && bst != LZF_OUTPUT_FULL) st = bst; }
            if (st != LZF_OK) { status[f] = st; continue; }
            size_t w = write_header(s, bd, out[f]);
            for (size_t i = 0; i < nb; ++i) {
                const size_t q = frame_jobs[f][i], off = i * bs, n = in_len[f] - off < bs ? in_len[f] - off : bs;
                const bool stored = res[q].status == LZF_OUTPUT_FULL;                                     // :250-255
                const uint32_t len = stored ? (uint32_t)n : (uint32_t)res[q].out_len;
                wr32(out[f] + w, stored ? (len | INCOMPRESSIBLE) : len); w += 4;                           // :247,253
                if (stored) stored_copies.push_back({out[f] + w, in[f] + off, n});
                else if (len) down.push_back({job_out_off[q], out[f] + w, len});                          // :258
                w += len;
                if (bsum) {                                                                                // :259-263
                    h_ptr.push_back(stored ? jobs[q].input + jobs[q].cursor : jobs[q].out); h_len.push_back(len);
                    sum_at.push_back({out[f] + w, n_hashed++}); w += 4;
                }
            }
            wr32(out[f] + w, 0); w += 4;                                                                   // :277 EndMark
            if (csum) { content_at[f] = out[f] + w; w += 4; }                                              // :279-281
            out_len[f] = w;
        }
        if (!h_ptr.empty()) {
            Lists& rl = sum_lists[g];
            const size_t ip = rl.add(h_ptr.data(), h_ptr.size() * sizeof h_ptr[0]), il = rl.add(h_len.data(), h_len.size() * sizeof h_len[0]), io = rl.add(nullptr, h_ptr.size() * sizeof(uint32_t));
            HIPOK(hipStreamWaitEvent(hs, done.e[g], 0));
            RCOK(rl.upload(sg, S_LISTS_G + (int)g, hs));
            RCOK(lzf_xxh32_batch(rl.ptr<const uint8_t*>(ip), rl.ptr<uint64_t>(il), rl.ptr<uint32_t>(io), (uint32_t)h_ptr.size(), hs));
            HIPOK(hipMemcpyAsync(mbox + mb_sums + sizeof(uint32_t) * hash0, rl.ptr<uint32_t>(io), h_ptr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
            g_stats.device_block_hashes += h_ptr.size();
        }
        TRACE("c:   group layout");
        HIPOK(sg.download(down, out_total, reinterpret_cast<const uint8_t*>(dout_b[g]), nullptr, done.e[g]));          // (returns when the group's payloads are in place)
        TRACE("c:   group download");
    }
    TRACE("c: downloads");
    sg.parallel_for(stored_copies.size(), [&](size_t i) { memcpy(stored_copies[i].dst, stored_copies[i].src, stored_copies[i].n); });
    HIPOK(hipStreamSynchronize(hs));
    HIPOK(hipStreamSynchronize(cs));
    TRACE("c: checksums home");
    const uint32_t* const sums = reinterpret_cast<const uint32_t*>(mbox + mb_sums);
    for (const Patch& pt : sum_at) wr32(pt.at, sums[pt.hash_index]);
    for (size_t i = 0; i < dev_hash_frames.size(); ++i) content[dev_hash_frames[i]] = chash[i];
    for (uint32_t f = 0; f < n_frames; ++f) if (content_at[f]) wr32(content_at[f], content[f]);
    ++g_stats.calls;
    return LZF_OK;
}

}  // extern "C"
namespace {

struct DFrame {
    lzf_frame_info fi; std::vector<Blk> blocks; FrameScan sc; bool live = false, linked = false;
    size_t in_off = 0, out_off = 0, out_size = 0;      // device offsets: the frame's bytes; linked: the stream's output buffer
    std::vector<size_t> job;                             // per block: job index or SIZE_MAX (stored)
    std::vector<size_t> slot;                            // independent: device offset of the block's output slot
    uint32_t chain = 0;                                  // linked: index among the linked streams
    size_t need = 0;                                     // device bytes this frame asks for (input + output room + packed result)
};
// the most a block of `len` compressed bytes can decode to: every byte a 255-run length byte (raw/decompress.rs:40-56)
inline size_t block_out_bound(size_t bmax, size_t len) { const size_t e = 255 * len + 16; return e < bmax ? e : bmax; }

// One pass over frames [f0, f1): everything on the device at once.
int decompress_group(Staging& sg, std::vector<DFrame>& fr, uint32_t f0, uint32_t f1, const uint8_t* const* in, const size_t* in_len,
                     const uint8_t* dict, size_t dict_len, uint8_t* const* out, const size_t* out_cap, size_t* out_len, size_t* consumed, int* status) {
    hipStream_t cs = sg.stream(0), hs = sg.stream(3);
    if (!cs || !hs) return fail_hip();
    TRACE_BEGIN();
    size_t in_total = 0, out_total = 0, max_steps = 0, n_sums = 0, pack_bound = 0;
    uint32_t n_chain = 0;
    std::vector<Seg> up;
    for (uint32_t f = f0; f < f1; ++f) {
        DFrame& F = fr[f];
        if (!F.live) continue;
        const size_t nb = F.blocks.size(), bmax = (size_t)F.fi.block_maxsize;
        F.in_off = in_total; in_total = up256(in_total + F.sc.consumed);
        if (F.sc.consumed) up.push_back({F.in_off, const_cast<uint8_t*>(in[f]), F.sc.consumed});
        F.job.assign(nb, SIZE_MAX); F.slot.assign(nb, 0);
        if (F.fi.flags & FL_BLOCKSUM) n_sums += nb;
        size_t bound = 0;
        for (size_t i = 0; i < nb; ++i) bound += F.blocks[i].compressed ? block_out_bound(bmax, F.blocks[i].len) : F.blocks[i].len;
        pack_bound = up256(pack_bound + bound);
        if (F.linked) {
            // a block may run past its limit by its literals before the stream is stopped: room for that
            if (nb) { F.chain = n_chain++; F.out_off = out_total; F.out_size = bound + F.sc.consumed + 64; out_total = up256(out_total + F.out_size); if (nb > max_steps) max_steps = nb; }
        } else {
            for (size_t i = 0; i < nb; ++i) if (F.blocks[i].compressed) { F.slot[i] = out_total; out_total = up256(out_total + block_out_bound(bmax, F.blocks[i].len) + F.blocks[i].len); }   // limit + C
        }
    }
    if (!sg.pinned(in_total > pack_bound ? in_total : pack_bound)) return fail_hip();
    uint8_t* const din = static_cast<uint8_t*>(sg.device(S_IN, in_total));
    uint8_t* const dout = static_cast<uint8_t*>(sg.device(S_OUT, out_total));
    uint8_t* d_dict = nullptr;
    if (!din || !dout) return fail_hip();
    if (dict_len) { d_dict = static_cast<uint8_t*>(sg.device(S_DICT, dict_len)); if (!d_dict) return fail_hip(); HIPOK(hipMemcpyAsync(d_dict, dict, dict_len, hipMemcpyHostToDevice, cs)); }
    // ---- job list ordered by step: step 0 = every block of the independent frames + block 0 of the linked streams
    std::vector<lzf_decompress_job> jobs;
    std::vector<size_t> step_off;
    const size_t n_steps = max_steps > 1 ? max_steps : 1;
    std::vector<lzf_chain_step> csteps((size_t)n_chain * n_steps);
    for (size_t k = 0; k < n_steps; ++k) {
        step_off.push_back(jobs.size());
        for (uint32_t f = f0; f < f1; ++f) {
            DFrame& F = fr[f];
            if (!F.live) continue;
            const size_t nb = F.blocks.size(), bmax = (size_t)F.fi.block_maxsize;
            auto add_job = [&](size_t i) {
                lzf_decompress_job j;
                memset(&j, 0, sizeof j);
                j.input = din + F.in_off + (F.blocks[i].data - in[f]); j.input_len = F.blocks[i].len;
                j.prefix = d_dict; j.prefix_len = dict_len;                                       // :239-245
                const size_t lim = bmax;                                                          // :248
                if (F.linked) { j.out = dout + F.out_off; j.out_cap = lim + F.blocks[i].len; j.output_limit = lim; }   // (patched per step)
                else { j.out = dout + F.slot[i]; j.out_cap = block_out_bound(bmax, F.blocks[i].len) + F.blocks[i].len; j.output_limit = lim; }
                F.job[i] = jobs.size(); jobs.push_back(j);
            };
            if (!F.linked) { if (k == 0) for (size_t i = 0; i < nb; ++i) if (F.blocks[i].compressed) add_job(i); continue; }
            if (!nb) continue;
            lzf_chain_step& st = csteps[k * n_chain + F.chain];
            memset(&st, 0, sizeof st);
            st.prev_job = (k > 0 && k - 1 < nb && F.blocks[k - 1].compressed) ? (uint32_t)F.job[k - 1] : UINT32_MAX;
            st.job = UINT32_MAX; st.out = dout + F.out_off; st.block_maxsize = bmax;
            if (k < nb) {
                if (F.blocks[k].compressed) { add_job(k); st.job = (uint32_t)F.job[k]; }
                else { st.stored_len = F.blocks[k].len; st.stored_src = din + F.in_off + (F.blocks[k].data - in[f]); }
            }
        }
    }
    step_off.push_back(jobs.size());
    const size_t n_jobs = jobs.size();
    if (n_jobs > 0x7FFFFFFFull) return LZF_E_INVALID;
    // results come back through the pinned mailbox: [block results | block checksums | content hashes]
    const size_t mb_sums = up256(sizeof(lzf_job_result) * n_jobs), mb_chash = mb_sums + up256(sizeof(uint32_t) * n_sums);
    uint8_t* const mbox = sg.mailbox(mb_chash + sizeof(uint32_t) * (f1 - f0));
    if (!mbox) return fail_hip();
    const lzf_job_result* const res = reinterpret_cast<const lzf_job_result*>(mbox);
    const uint32_t* const sums = reinterpret_cast<const uint32_t*>(mbox + mb_sums);
    const uint32_t* const chash = reinterpret_cast<const uint32_t*>(mbox + mb_chash);
    TRACE("d: layout + scratch");
    HIPOK(sg.upload(up, in_total, din));
    TRACE("d: upload issued");
    // ---- block checksums (:228-235): every block of every frame that carries them, one launch on the second stream
    Lists bl;
    if (n_sums) {
        std::vector<const uint8_t*> p; std::vector<uint64_t> n;
        for (uint32_t f = f0; f < f1; ++f) if (fr[f].live && (fr[f].fi.flags & FL_BLOCKSUM))
            for (const Blk& b : fr[f].blocks) { p.push_back(din + fr[f].in_off + (b.data - in[f])); n.push_back(b.len); }
        const size_t ip = bl.add(p.data(), p.size() * sizeof p[0]), il = bl.add(n.data(), n.size() * sizeof n[0]), io = bl.add(nullptr, n_sums * sizeof(uint32_t));
        HIPOK(sg.join_copies(hs));
        RCOK(bl.upload(sg, S_HASH, hs));
        RCOK(lzf_xxh32_batch(bl.ptr<const uint8_t*>(ip), bl.ptr<uint64_t>(il), bl.ptr<uint32_t>(io), (uint32_t)n_sums, hs));
        HIPOK(hipMemcpyAsync(mbox + mb_sums, bl.ptr<uint32_t>(io), n_sums * sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
        g_stats.device_block_hashes += n_sums;
    }
    if (n_jobs || n_chain) {
        lzf_decompress_job* const d_jobs = static_cast<lzf_decompress_job*>(sg.device(S_JOBS, sizeof(lzf_decompress_job) * n_jobs));
        lzf_job_result* const d_res = static_cast<lzf_job_result*>(sg.device(S_RES, sizeof(lzf_job_result) * n_jobs));
        if (!d_jobs || !d_res) return fail_hip();
        if (n_jobs) HIPOK(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(lzf_decompress_job) * n_jobs, hipMemcpyHostToDevice, cs));
        lzf_chain_step* d_steps = nullptr; lzf_chain_state* d_state = nullptr;
        if (n_chain) {
            d_steps = static_cast<lzf_chain_step*>(sg.device(S_STEPS, sizeof(lzf_chain_step) * csteps.size()));
            d_state = static_cast<lzf_chain_state*>(sg.device(S_STATE, sizeof(lzf_chain_state) * n_chain));
            if (!d_steps || !d_state) return fail_hip();
            HIPOK(hipMemcpyAsync(d_steps, csteps.data(), sizeof(lzf_chain_step) * csteps.size(), hipMemcpyHostToDevice, cs));
            HIPOK(hipMemsetAsync(d_state, 0, sizeof(lzf_chain_state) * n_chain, cs));
        }
        HIPOK(sg.join_copies(cs));
        for (size_t k = 0; k < n_steps; ++k) {
            if (n_chain) RCOK(lzf_chain_decompress_step(d_steps + k * n_chain, d_state, n_chain, d_jobs, d_res, cs));
            const size_t a = step_off[k], cnt = step_off[k + 1] - a;
            if (cnt) {
                uint64_t max_in = 0;                                                       // (the host built the jobs: it knows their sizes)
                for (size_t q = a; q < a + cnt; ++q) if (jobs[q].input_len > max_in) max_in = jobs[q].input_len;
                RCOK(lzf_decompress_batch_sized(d_jobs + a, d_res + a, (uint32_t)cnt, max_in, cs));
            }
        }
        if (n_jobs) HIPOK(hipMemcpyAsync(mbox, d_res, sizeof(lzf_job_result) * n_jobs, hipMemcpyDeviceToHost, cs));
    } else {
        HIPOK(sg.join_copies(cs));
    }
    TRACE("d: launches");
    HIPOK(hipStreamSynchronize(cs));
    HIPOK(hipStreamSynchronize(hs));
    TRACE("d: kernels done");
    // ---- per frame: the delivery loop of the reader (decompress.rs:198-288), on the block lengths the device reports
    std::vector<const uint8_t*> r_src; std::vector<uint64_t> r_len; std::vector<size_t> pk_pos;
    std::vector<Seg> down;
    struct Want { uint32_t f; size_t pk; size_t n; };
    std::vector<Want> dev_hash, host_hash;
    size_t pk_total = 0, sum_i = 0; uint64_t r_max = 0;
    for (uint32_t f = f0; f < f1; ++f) {
        DFrame& F = fr[f];
        if (!F.live) continue;
        const size_t nb = F.blocks.size(), bmax = (size_t)F.fi.block_maxsize;
        const bool csum = F.fi.flags & FL_CSUM, bsum = F.fi.flags & FL_BLOCKSUM;
        const size_t sum0 = sum_i; if (bsum) sum_i += nb;
        size_t w = 0, hist = 0;        // bytes delivered; linked: length of the stream's device buffer so far
        int st = LZF_OK; bool stopped = false;
        const size_t pk0 = pk_total;
        auto range = [&](const uint8_t* src, size_t n) {
            if (!n) return;
            if (!r_src.empty() && r_src.back() + r_len.back() == src && pk_pos.back() + r_len.back() == pk0 + w) r_len.back() += n;   // (linked streams: one range)
            else { r_src.push_back(src); r_len.push_back(n); pk_pos.push_back(pk0 + w); }
            if (r_len.back() > r_max) r_max = r_len.back();
        };
        for (size_t i = 0; i < nb; ++i) {
            const Blk& b = F.blocks[i];
            if (consumed) consumed[f] = b.end_off;                                             // what the reader has read when it stops in this block
            if (bsum && sums[sum0 + i] != b.want_sum) { st = LZF_F_BLOCK_CHECKSUM_FAIL; break; }   // :228-235
            size_t n; const uint8_t* dsrc;                                                    // device address of the block's bytes
            if (b.compressed) {
                const lzf_job_result& r = res[F.job[i]];
                if (r.status != LZF_OK) { st = r.status; break; }                             // CodecError
                if (F.linked) { n = (size_t)r.out_len - hist; dsrc = dout + F.out_off + hist; }
                else { n = (size_t)r.out_len; dsrc = dout + F.slot[i]; }
            } else { n = b.len; dsrc = F.linked ? dout + F.out_off + hist : din + F.in_off + (b.data - in[f]); }   // :250 stored
            hist += n;
            if (n > bmax) { st = LZF_F_BLOCK_SIZE_OVERFLOW; break; }                          // :272-274
            if (out_cap[f] - w < n) { st = LZF_OUT_CAPACITY; break; }
            range(dsrc, n);
            w += n;
            if (n == 0) { stopped = true; break; }                                            // the io::Read adapter stops at an empty block (:52-71,:286)
        }
        out_len[f] = w;
        if (w) down.push_back({pk0, out[f], w});
        pk_total = up256(pk0 + w);
        if (st != LZF_OK) { status[f] = st; continue; }
        if (stopped) continue;
        if (consumed) consumed[f] = F.sc.consumed;
        if (F.sc.err != LZF_OK) { status[f] = F.sc.err; continue; }
        if (F.sc.endmark && csum) { (w <= kDeviceHashMax ? dev_hash : host_hash).push_back({f, pk0, w}); }   // :207-211
    }
    if (!r_src.empty() || !dev_hash.empty()) {
        uint8_t* const dpack = static_cast<uint8_t*>(sg.device(S_PACK, pk_total));
        if (!dpack) return fail_hip();
        std::vector<uint8_t*> r_dst(r_src.size());
        for (size_t i = 0; i < r_src.size(); ++i) r_dst[i] = dpack + pk_pos[i];
        std::vector<const uint8_t*> hp; std::vector<uint64_t> hn;
        for (const Want& h : dev_hash) { hp.push_back(dpack + h.pk); hn.push_back(h.n); }
        Lists rl;
        const size_t is = rl.add(r_src.data(), r_src.size() * sizeof(void*)), id = rl.add(r_dst.data(), r_dst.size() * sizeof(void*)), il = rl.add(r_len.data(), r_len.size() * sizeof(uint64_t));
        const size_t ihp = rl.add(hp.data(), hp.size() * sizeof(void*)), ihn = rl.add(hn.data(), hn.size() * sizeof(uint64_t)), iho = rl.add(nullptr, hp.size() * sizeof(uint32_t));
        RCOK(rl.upload(sg, S_LISTS, cs));
        RCOK(lzf_copy_ranges(rl.ptr<const uint8_t*>(is), rl.ptr<uint8_t*>(id), rl.ptr<uint64_t>(il), (uint32_t)r_src.size(), r_max, cs));
        if (!hp.empty()) {                                                                    // the content hash runs on the second stream while the output goes home
            hipEvent_t packed = nullptr;
            HIPOK(hipEventCreateWithFlags(&packed, hipEventDisableTiming));
            HIPOK(hipEventRecord(packed, cs)); HIPOK(hipStreamWaitEvent(hs, packed, 0));
            RCOK(lzf_xxh32_batch(rl.ptr<const uint8_t*>(ihp), rl.ptr<uint64_t>(ihn), rl.ptr<uint32_t>(iho), (uint32_t)hp.size(), hs));
            HIPOK(hipMemcpyAsync(mbox + mb_chash, rl.ptr<uint32_t>(iho), hp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
            HIPOK(hipEventDestroy(packed));
            g_stats.device_content_hashes += hp.size();
        }
        TRACE("d: delivery + pack issued");
        HIPOK(sg.download(down, pk_total, dpack, cs));
        TRACE("d: download");
        HIPOK(hipStreamSynchronize(hs));
        TRACE("d: content hashes");
    }
    for (size_t i = 0; i < dev_hash.size(); ++i) if (fr[dev_hash[i].f].sc.want_content != chash[i]) status[dev_hash[i].f] = LZF_F_FRAME_CHECKSUM_FAIL;
    if (!host_hash.empty()) {
        std::vector<const uint8_t*> p; std::vector<size_t> n; std::vector<uint32_t> h;
        for (const Want& q : host_hash) { p.push_back(out[q.f]); n.push_back(q.n); }
        host_hashes(sg, p, n, h);
        for (size_t i = 0; i < h.size(); ++i) if (fr[host_hash[i].f].sc.want_content != h[i]) status[host_hash[i].f] = LZF_F_FRAME_CHECKSUM_FAIL;
    }
    return LZF_OK;
}
}  // namespace
extern "C" {

int lzf_frame_decompress_many(uint32_t n_frames, const uint8_t* const* in, const size_t* in_len, const uint8_t* dict, size_t dict_len,
                              uint8_t* const* out, const size_t* out_cap, size_t* out_len, size_t* consumed, int* status) {
    if (n_frames && (!in || !in_len || !out || !out_cap || !out_len || !status)) return LZF_E_INVALID;
    if (!dict) dict_len = 0;
    std::vector<DFrame> fr(n_frames);
    for (uint32_t f = 0; f < n_frames; ++f) {
        DFrame& F = fr[f];
        out_len[f] = 0; if (consumed) consumed[f] = 0;
        size_t hdr_read = 0;
        const int rc = read_header_ex(in[f], in_len[f], &F.fi, &hdr_read);
        if (rc != LZF_OK) { status[f] = rc; if (consumed) consumed[f] = hdr_read; continue; }
        status[f] = LZF_OK; F.live = true; F.linked = !(F.fi.flags & FL_INDEP);
        scan_blocks(in[f], in_len[f], F.fi, F.blocks, F.sc);
        if (consumed) consumed[f] = F.sc.consumed;
        // device memory the frame asks for: its bytes, an output slot per block bounded by what the block can expand to
        // (a 5-byte block cannot claim block_maxsize), and the packed result
        const size_t bmax = (size_t)F.fi.block_maxsize;
        size_t bound = 0;
        for (const Blk& b : F.blocks) bound += (b.compressed ? block_out_bound(bmax, b.len) : 0) + b.len + 256;
        F.need = F.sc.consumed + 2 * bound + 4096;
    }
    if (n_frames == 0) return LZF_OK;
    Staging& sg = Staging::get();
    std::lock_guard<std::mutex> guard(sg.lock());
    // ---- slices: as many frames per pass as the memory budget holds (half of what the device has free; pinned host memory is
    //      the staging's business: a ring of at most 2 GiB); a frame that does not fit the budget alone reports LZF_E_NO_MEMORY
    size_t free_b = 0, total_b = 0;
    HIPOK(hipMemGetInfo(&free_b, &total_b));
    size_t budget = g_budget ? g_budget : free_b / 2;
    for (uint32_t f0 = 0; f0 < n_frames;) {
        size_t sum = 0; uint32_t f1 = f0;
        while (f1 < n_frames && (f1 == f0 || sum + fr[f1].need <= budget)) {
            sum += fr[f1].live ? fr[f1].need : 0; ++f1;
        }
        if (f1 == f0 + 1 && fr[f0].live && fr[f0].need > budget) {
            status[f0] = LZF_E_NO_MEMORY; out_len[f0] = 0; if (consumed) consumed[f0] = 0;
            f0 = f1; continue;
        }
        RCOK(decompress_group(sg, fr, f0, f1, in, in_len, dict, dict_len, out, out_cap, out_len, consumed, status));
        f0 = f1;
    }
    ++g_stats.calls;
    return LZF_OK;
}

void lzf_frame_set_memory_budget(size_t bytes) {
    Staging& sg = Staging::get();
    std::lock_guard<std::mutex> g(sg.lock());
    g_budget = bytes;
}

// ---- streaming frame writer (compress.rs:138-157, :160-282 with the stream fed piece by piece) ------------------------------
struct lzf_frame_writer {
    lzf_settings s{};
    std::vector<uint8_t> dict;
    uint8_t bd = 0;
    lzf_write_all_fn sink = nullptr; void* ctx = nullptr;
    size_t bs = 0; uint32_t per_launch = 64;
    std::vector<uint8_t> pending;          // stream bytes not compressed yet
    std::vector<uint8_t> in_buffer;        // linked blocks: the reference's in_buffer (window ++ block), compress.rs:217-222
    lzf_u32_table templ{}, table{};        // template_table / table (:202,:220)
    bool seeded = false, header_done = false, dead = false, finished = false;
    int sink_err = 0;
    Xxh32 content{0};
};

namespace {
int fw_put(lzf_frame_writer* w, const uint8_t* p, size_t n) {
    if (n == 0) return LZF_OK;
    const int e = w->sink(w->ctx, p, n);
    if (e != 0) { w->sink_err = e; w->dead = true; return LZF_OUTPUT_FULL; }
    return LZF_OK;
}
int fw_put32(lzf_frame_writer* w, uint32_t v) { uint8_t b[4]; wr32(b, v); return fw_put(w, b, 4); }
int fw_header(lzf_frame_writer* w) {
    if (w->header_done) return LZF_OK;
    uint8_t h[32];
    const size_t n = write_header(&w->s, w->bd, h);
    w->header_done = true;
    return fw_put(w, h, n);                                                     // :200 writer.write_all(&header)
}
// length word, payload, optional checksum of one block (:244-263)
int fw_emit_block(lzf_frame_writer* w, const uint8_t* payload, uint32_t n, bool stored, uint32_t sum) {
    int rc = fw_put32(w, stored ? (n | INCOMPRESSIBLE) : n);
    if (rc == LZF_OK) rc = fw_put(w, payload, n);
    if (rc == LZF_OK && w->s.block_checksums) rc = fw_put32(w, sum);
    return rc;
}
// block checksums of a batch on the device (one launch), like the other drivers
int fw_sums(const std::vector<const uint8_t*>& p, const std::vector<uint64_t>& n, std::vector<uint32_t>& out) {
    out.assign(p.size(), 0);
    if (p.empty()) return LZF_OK;
    g_reader_device_hashes += p.size();
    return lzf_xxh32_batch_host(p.data(), n.data(), out.data(), (uint32_t)p.size());
}
// independent blocks: the first `nblk` blocks of pending (the last may be short) in one launch
int fw_flush_independent(lzf_frame_writer* w, size_t nblk, size_t bytes) {
    const size_t dl = w->dict.size();
    std::vector<lzf_compress_job> jobs(nblk);
    std::vector<lzf_job_result> res(nblk);
    std::vector<uint8_t> inbuf, outbuf(bytes);
    if (dl) inbuf.resize(nblk * dl + bytes);
    size_t off = 0, ioff = 0;
    for (size_t i = 0; i < nblk; ++i) {
        const size_t n = bytes - off < w->bs ? bytes - off : w->bs;
        lzf_compress_job& j = jobs[i];
        memset(&j, 0, sizeof j);
        if (dl) {                                                               // in_buffer = dictionary ++ block (:218-222,:268)
            memcpy(inbuf.data() + ioff, w->dict.data(), dl);
            memcpy(inbuf.data() + ioff + dl, w->pending.data() + off, n);
            j.input = inbuf.data() + ioff; j.input_len = dl + n; j.cursor = dl; ioff += dl + n;
            if (w->seeded) { j.table = &w->templ; j.flags = LZF_CJOB_TABLE_READONLY; }      // table = template_table.clone() (:220,:270)
        } else { j.input = w->pending.data() + off; j.input_len = n; j.cursor = 0; }
        j.out = outbuf.data() + off; j.out_cap = n;                             // NoPartialWrites(&mut out_buffer[..read_bytes]) (:242)
        j.table_kind = LZF_TABLE_U32;
        off += n;
    }
    int rc = lzf_compress_batch_host(jobs.data(), res.data(), (uint32_t)nblk);
    if (rc != LZF_OK) return rc;
    std::vector<const uint8_t*> pp(nblk); std::vector<uint64_t> pn(nblk); std::vector<uint32_t> sums;
    off = 0;
    for (size_t i = 0; i < nblk; ++i) {
        const size_t n = bytes - off < w->bs ? bytes - off : w->bs;
        if (res[i].status != LZF_OK && res[i].status != LZF_OUTPUT_FULL) return res[i].status;
        const bool stored = res[i].status == LZF_OUTPUT_FULL;                  // :250-255
        pp[i] = stored ? w->pending.data() + off : outbuf.data() + off;
        pn[i] = stored ? n : res[i].out_len;
        off += n;
    }
    if (w->s.block_checksums) { rc = fw_sums(pp, pn, sums); if (rc != LZF_OK) return rc; }
    off = 0;
    for (size_t i = 0; i < nblk; ++i) {
        const size_t n = bytes - off < w->bs ? bytes - off : w->bs;
        if (w->s.content_checksum) w->content.update(w->pending.data() + off, n);          // :233-235
        rc = fw_emit_block(w, pp[i], (uint32_t)pn[i], res[i].status == LZF_OUTPUT_FULL, w->s.block_checksums ? sums[i] : 0);
        if (rc != LZF_OK) return rc;
        off += n;
    }
    w->pending.erase(w->pending.begin(), w->pending.begin() + (ptrdiff_t)bytes);
    return LZF_OK;
}
// linked blocks: one block (n bytes of pending) behind the carried window, the table carried (:221-275)
int fw_flush_linked(lzf_frame_writer* w, size_t n) {
    const size_t window_offset = w->in_buffer.size();
    w->in_buffer.insert(w->in_buffer.end(), w->pending.begin(), w->pending.begin() + (ptrdiff_t)n);
    if (w->s.content_checksum) w->content.update(w->in_buffer.data() + window_offset, n);
    std::vector<uint8_t> outbuf(n);
    lzf_compress_job j; memset(&j, 0, sizeof j);
    j.input = w->in_buffer.data(); j.input_len = w->in_buffer.size(); j.cursor = window_offset;
    j.out = outbuf.data(); j.out_cap = n; j.table = &w->table; j.table_kind = LZF_TABLE_U32;
    lzf_job_result r{};
    int rc = lzf_compress_batch_host(&j, &r, 1);
    if (rc != LZF_OK) return rc;
    if (r.status != LZF_OK && r.status != LZF_OUTPUT_FULL) return r.status;
    const bool stored = r.status == LZF_OUTPUT_FULL;
    const uint8_t* payload = stored ? w->in_buffer.data() + window_offset : outbuf.data();
    const uint32_t plen = stored ? (uint32_t)n : (uint32_t)r.out_len;
    uint32_t sum = 0;
    if (w->s.block_checksums) { std::vector<const uint8_t*> pp{payload}; std::vector<uint64_t> pn{plen}; std::vector<uint32_t> so; rc = fw_sums(pp, pn, so); if (rc != LZF_OK) return rc; sum = so[0]; }
    rc = fw_emit_block(w, payload, plen, stored, sum);
    if (rc != LZF_OK) return rc;
    if (w->in_buffer.size() > LZF_WINDOW_SIZE) {                                // :271-275
        const size_t forget = w->in_buffer.size() - LZF_WINDOW_SIZE;
        w->table.offset += forget;
        w->in_buffer.erase(w->in_buffer.begin(), w->in_buffer.begin() + (ptrdiff_t)forget);
    }
    w->pending.erase(w->pending.begin(), w->pending.begin() + (ptrdiff_t)n);
    return LZF_OK;
}
// compress what is buffered: whole blocks only, or (final) everything
int fw_drain(lzf_frame_writer* w, bool final) {
    for (;;) {
        const size_t have = w->pending.size();
        if (have == 0) return LZF_OK;
        int rc;
        if (w->s.independent_blocks) {
            size_t nfull = have / w->bs;
            if (!final && nfull < w->per_launch) return LZF_OK;
            size_t nblk = nfull < w->per_launch ? nfull : w->per_launch, bytes = nblk * w->bs;
            if (final && nblk < w->per_launch && have > bytes) { ++nblk; bytes = have; }    // the short last block rides along
            if (nblk == 0) return LZF_OK;
            rc = fw_flush_independent(w, nblk, bytes);
        } else {
            if (!final && have < w->bs) return LZF_OK;
            rc = fw_flush_linked(w, have < w->bs ? have : w->bs);
        }
        if (rc != LZF_OK) { w->dead = true; return rc; }
    }
}
}  // namespace

int lzf_frame_writer_new(const lzf_settings* s, lzf_write_all_fn write_all, void* ctx, uint32_t blocks_per_launch, lzf_frame_writer** out) {
    if (!s || !write_all || !out) return LZF_E_INVALID;
    *out = nullptr;
    uint8_t bd;
    int rc = bd_new(s->block_size, &bd);                                         // :183
    if (rc != LZF_OK) return rc;
    lzf_frame_writer* w = new lzf_frame_writer();
    w->s = *s; w->bd = bd; w->sink = write_all; w->ctx = ctx; w->bs = (size_t)s->block_size;
    w->per_launch = blocks_per_launch ? blocks_per_launch : 64u;
    if (s->dictionary && s->dictionary_len) {
        w->dict.assign(s->dictionary, s->dictionary + s->dictionary_len);
        rc = seeded_template(w->dict.data(), w->dict.size(), &w->templ);         // :202-214
        if (rc != LZF_OK) { delete w; return rc; }
        w->seeded = w->dict.size() >= 8;
    }
    w->s.dictionary = w->dict.empty() ? nullptr : w->dict.data();
    w->table = w->templ;                                                         // :220
    w->in_buffer = w->dict;                                                      // :218 in_buffer.extend_from_slice(block_initializer)
    *out = w;
    return LZF_OK;
}

int lzf_frame_writer_write(lzf_frame_writer* w, const uint8_t* data, size_t len) {
    if (!w || (!data && len)) return LZF_E_INVALID;
    if (w->dead || w->finished) return w->dead && w->sink_err ? LZF_OUTPUT_FULL : LZF_E_INVALID;
    int rc = fw_header(w);
    if (rc != LZF_OK) return rc;
    // (whole launches' worth is compressed as soon as it is there, so `pending` never holds more than one launch + one feed)
    size_t done = 0;
    while (done < len) {
        const size_t room = w->s.independent_blocks ? (size_t)w->per_launch * w->bs : w->bs;
        size_t take = len - done;
        if (w->pending.size() < room && take > room - w->pending.size()) take = room - w->pending.size();
        w->pending.insert(w->pending.end(), data + done, data + done + take);
        done += take;
        rc = fw_drain(w, false);
        if (rc != LZF_OK) return rc;
    }
    return LZF_OK;
}

int lzf_frame_writer_finish(lzf_frame_writer* w) {
    if (!w) return LZF_E_INVALID;
    if (w->dead || w->finished) return w->dead && w->sink_err ? LZF_OUTPUT_FULL : LZF_E_INVALID;
    int rc = fw_header(w);
    if (rc == LZF_OK) rc = fw_drain(w, true);
    if (rc == LZF_OK) rc = fw_put32(w, 0);                                       // :277 EndMark
    if (rc == LZF_OK && w->s.content_checksum) rc = fw_put32(w, w->content.digest());      // :279-281
    w->finished = true;
    return rc;
}

int lzf_frame_writer_sink_error(const lzf_frame_writer* w) { return w ? w->sink_err : 0; }
void lzf_frame_writer_free(lzf_frame_writer* w) { delete w; }

}  // extern "C"
