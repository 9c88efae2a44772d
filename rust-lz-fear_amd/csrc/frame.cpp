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
#include "../../include/lzfear_frame.h"

namespace {

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
        return LZF_E_HIP;
    }
    int rc = LZF_OK;
    if (hipMemcpy(d_dict, dict, dict_len, hipMemcpyHostToDevice) != hipSuccess) rc = LZF_E_HIP;
    if (rc == LZF_OK) rc = lzf_table_seed_from_dictionary((lzf_u32_table*)d_tab, (const uint8_t*)d_dict, dict_len, nullptr);
    if (rc == LZF_OK && hipMemcpy(host_table, d_tab, sizeof *host_table, hipMemcpyDeviceToHost) != hipSuccess) rc = LZF_E_HIP;
    (void)hipFree(d_dict); (void)hipFree(d_tab);
    return rc;
}

// One block of a frame as the scan finds it (decompress.rs:217-235).
struct Blk { const uint8_t* data; uint32_t len; bool compressed; };
struct FrameScan {
    size_t consumed = 0;          // bytes of the input read
    int err = LZF_OK;             // structural error that ends the scan (reported in stream order)
    bool endmark = false;
    uint32_t want_content = 0;    // content checksum behind the EndMark
};
// The u32 length hops over a frame's blocks (decompress.rs:205-235), block checksums verified on the way.
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
        if (bsum) {                                                                           // :228-235
            if (in_len - r < 4) { sc.err = LZF_F_INPUT_ERROR; r = in_len; break; }
            const uint32_t c = rd32(in + r); r += 4;
            Xxh32 h; h.update(data, bl);
            if (c != h.digest()) { sc.err = LZF_F_BLOCK_CHECKSUM_FAIL; break; }
        }
        blocks.push_back({data, bl, compressed});
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
        if (s->block_checksums) { wr32(out + w + n, lzf_xxh32(out + w, n, 0)); }   // :259-263
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

// decompress.rs:102-161
int lzf_frame_read_header(const uint8_t* in, size_t in_len, lzf_frame_info* info) {
    memset(info, 0, sizeof *info);
    size_t r = 0;
#define NEED(n) do { if (in_len - r < (size_t)(n)) return LZF_F_INPUT_ERROR; } while (0)
    NEED(4); if (rd32(in) != LZF_MAGIC) return LZF_F_WRONG_MAGIC; r = 4;     // :103-106
    NEED(1); const uint8_t flg = in[r++];
    if ((flg >> 6) != 1) return LZF_F_UNSUPPORTED_VERSION;                    // header.rs:33-36
    if (flg & 0x02) return LZF_F_RESERVED_FLAG_BITS;                          // header.rs:37-39
    NEED(1); const uint8_t bd = in[r++];
    if (bd & 0x8F) return LZF_F_RESERVED_BD_BITS;                             // header.rs:66-68
    info->flags = flg; info->bd = bd;
    if (flg & FL_CSIZE) { NEED(8); info->has_content_size = 1; info->content_size = (uint64_t)rd32(in + r) | ((uint64_t)rd32(in + r + 4) << 32); r += 8; }
    if (flg & FL_DICTID) { NEED(4); info->has_dictionary_id = 1; info->dictionary_id = rd32(in + r); r += 4; }
    NEED(1); const uint8_t hc = in[r++];
    if (hc != (uint8_t)(lzf_xxh32(in + 4, r - 5, 0) >> 8)) return LZF_F_HEADER_CHECKSUM_FAIL;   // :132-136
    const unsigned size = (bd >> 4) & 7;
    if (size < 4) return LZF_F_UNIMPLEMENTED_BLOCKSIZE;                       // :153, header.rs:73-80
    info->block_maxsize = 1ull << (size * 2 + 8);
    info->header_len = (uint16_t)r;
    return LZF_OK;
#undef NEED
}

// decompress.rs:198-288 — one frame = a batch of one (lzf_frame_decompress_many below)
int lzf_frame_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len,
                         uint8_t* out, size_t out_cap, size_t* out_len, size_t* consumed) {
    *out_len = 0; if (consumed) *consumed = 0;
    int st = LZF_OK;
    const int rc = lzf_frame_decompress_many(1, &in, &in_len, dict, dict_len, &out, &out_cap, out_len, consumed, &st);
    return rc != LZF_OK ? rc : st;
}

// =====================================================================================================================
// Many frames per call.  One frame of a few large blocks leaves the chip almost empty (one wavefront per block); the
// batch entry points want thousands of blocks.  These drivers put every block of every frame into the same launches
// and keep inputs, tables and outputs on the device from the first block to the last: independent-block frames are
// one launch; linked-block frames advance together, block k of every stream in launch k, with no host round trip in
// between (the jobs of all steps are known up front on the compress side — the window is input data and the table
// lives on the device; on the decompress side lzf_chain_decompress_step patches each stream's length on the device).
// Same bytes and the same statuses as calling lzf_frame_compress / lzf_frame_decompress once per frame.
// =====================================================================================================================
}  // extern "C"
namespace {
struct DBuf {
    void* p = nullptr;
    ~DBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, n ? n : 1) == hipSuccess ? LZF_OK : LZF_E_HIP; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
#define HIPOK(e) do { if ((e) != hipSuccess) return LZF_E_HIP; } while (0)
}  // namespace
extern "C" {

int lzf_frame_compress_many(const lzf_settings* s, uint32_t n_frames, const uint8_t* const* in, const size_t* in_len,
                            uint8_t* const* out, const size_t* out_cap, size_t* out_len, int* status) {
    if (!s || (n_frames && (!in || !in_len || !out || !out_cap || !out_len || !status))) return LZF_E_INVALID;
    uint8_t bd;
    const int bd_rc = bd_new(s->block_size, &bd);                              // compress.rs:183
    for (uint32_t f = 0; f < n_frames; ++f) { out_len[f] = 0; status[f] = bd_rc != LZF_OK ? bd_rc : out_cap[f] < lzf_frame_compress_bound(s, in_len[f]) ? LZF_OUT_CAPACITY : LZF_OK; }
    if (bd_rc != LZF_OK || n_frames == 0) return LZF_OK;
    const size_t bs = (size_t)s->block_size;
    const uint8_t* dict = s->dictionary;
    const size_t dict_len = dict ? (size_t)s->dictionary_len : 0;
    const bool indep = s->independent_blocks != 0;
    const bool per_block_prefix = indep && dict_len > 0;                       // :218,:268 in_buffer = dict ++ block, for every block

    // ---- layout: input slab (host staging -> one copy), output slab (cap n per block, :242), job list ordered by step
    struct Fr { size_t nb, in_off, job0; };
    std::vector<Fr> fr(n_frames);
    size_t in_total = 0, out_total = 0, n_jobs = 0, max_nb = 0;
    for (uint32_t f = 0; f < n_frames; ++f) {
        fr[f].nb = status[f] == LZF_OK ? (in_len[f] + bs - 1) / bs : 0;
        fr[f].in_off = in_total;
        if (fr[f].nb) in_total = up256(in_total + (per_block_prefix ? fr[f].nb * dict_len : dict_len) + in_len[f]);
        n_jobs += fr[f].nb; if (fr[f].nb > max_nb) max_nb = fr[f].nb;
    }
    std::vector<lzf_compress_job> jobs(n_jobs ? n_jobs : 1);
    std::vector<uint32_t> job_frame(n_jobs), job_block(n_jobs);
    std::vector<size_t> job_out_off(n_jobs);
    std::vector<size_t> step_off;                                               // linked: jobs of step k are [step_off[k], step_off[k+1])
    std::vector<uint8_t> h_in(in_total ? in_total : 1);
    DBuf d_in, d_out, d_jobs, d_res, d_tabs, d_tabptr, d_adds, d_tmpl;
    int rc = d_in.alloc(in_total);
    if (rc != LZF_OK) return rc;
    uint8_t* const din = d_in.as<uint8_t>();

    lzf_u32_table tmpl;
    if (dict_len >= 8) { rc = seeded_template(dict, dict_len, &tmpl); if (rc != LZF_OK) return rc; } else memset(&tmpl, 0, sizeof tmpl);
    std::vector<void*> h_tabptr(n_jobs ? n_jobs : 1, nullptr);
    std::vector<uint64_t> h_adds(n_jobs ? n_jobs : 1, 0);
    uint32_t n_linked = 0;
    std::vector<uint32_t> lf_index(n_frames, 0);
    if (!indep) for (uint32_t f = 0; f < n_frames; ++f) if (fr[f].nb) lf_index[f] = n_linked++;
    if (!indep) { rc = d_tabs.alloc(sizeof(lzf_u32_table) * (size_t)(n_linked ? n_linked : 1)); if (rc != LZF_OK) return rc; }
    else if (dict_len >= 8) { rc = d_tmpl.alloc(sizeof tmpl); if (rc != LZF_OK) return rc; HIPOK(hipMemcpy(d_tmpl.p, &tmpl, sizeof tmpl, hipMemcpyHostToDevice)); }

    size_t jn = 0;
    if (indep) {
        for (uint32_t f = 0; f < n_frames; ++f) {
            fr[f].job0 = jn;
            size_t w = fr[f].in_off;
            if (fr[f].nb && !per_block_prefix) { memcpy(h_in.data() + w, in[f], in_len[f]); }
            for (size_t i = 0; i < fr[f].nb; ++i, ++jn) {
                const size_t off = i * bs, n = in_len[f] - off < bs ? in_len[f] - off : bs;
                lzf_compress_job& j = jobs[jn];
                memset(&j, 0, sizeof j);
                if (per_block_prefix) {
                    memcpy(h_in.data() + w, dict, dict_len); memcpy(h_in.data() + w + dict_len, in[f] + off, n);
                    j.input = din + w; j.input_len = dict_len + n; j.cursor = dict_len; w += dict_len + n;
                } else { j.input = din + fr[f].in_off + off; j.input_len = n; j.cursor = 0; }
                j.out_cap = n; j.table_kind = LZF_TABLE_U32;                   // :242, :202
                if (dict_len >= 8) { j.table = d_tmpl.p; j.flags = LZF_CJOB_TABLE_READONLY; }     // :220,:270 template.clone()
                job_frame[jn] = f; job_block[jn] = (uint32_t)i; job_out_off[jn] = out_total; out_total += n;
            }
        }
        step_off = {0, n_jobs};
    } else {
        // linked blocks (:271-275): in_buffer = the last <= 64 KiB of (dict ++ data so far) ++ block, a pointer into the
        // stream's slab; the table's offset grows by what the window forgets
        struct Ls { size_t lo, len; };                                          // in_buffer = slab[lo, lo + len)
        std::vector<Ls> ls(n_frames, Ls{0, dict_len});
        for (uint32_t f = 0; f < n_frames; ++f) if (fr[f].nb) {
            if (dict_len) memcpy(h_in.data() + fr[f].in_off, dict, dict_len);
            memcpy(h_in.data() + fr[f].in_off + dict_len, in[f], in_len[f]);
        }
        std::vector<uint64_t> pending_add(n_frames, 0);
        for (size_t k = 0; k < max_nb; ++k) {
            step_off.push_back(jn);
            for (uint32_t f = 0; f < n_frames; ++f) {
                if (k >= fr[f].nb) continue;
                const size_t off = k * bs, n = in_len[f] - off < bs ? in_len[f] - off : bs;
                lzf_compress_job& j = jobs[jn];
                memset(&j, 0, sizeof j);
                j.input = din + fr[f].in_off + ls[f].lo; j.input_len = ls[f].len + n; j.cursor = ls[f].len;   // :222,:243
                j.out_cap = n; j.table_kind = LZF_TABLE_U32;
                j.table = d_tabs.as<lzf_u32_table>() + lf_index[f];
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
    if (n_jobs == 0) {                                                          // only empty inputs: header + EndMark each
        for (uint32_t f = 0; f < n_frames; ++f) if (status[f] == LZF_OK) {
            uint32_t content = 0; if (s->content_checksum) content = lzf_xxh32(in[f], 0, 0);
            status[f] = lzf_frame_assemble(s, 0, nullptr, nullptr, nullptr, content, out[f], out_cap[f], &out_len[f]);
        }
        return LZF_OK;
    }
    rc = d_out.alloc(out_total); if (rc != LZF_OK) return rc;
    for (size_t q = 0; q < n_jobs; ++q) jobs[q].out = d_out.as<uint8_t>() + job_out_off[q];
    rc = d_jobs.alloc(sizeof(lzf_compress_job) * n_jobs); if (rc != LZF_OK) return rc;
    rc = d_res.alloc(sizeof(lzf_job_result) * n_jobs); if (rc != LZF_OK) return rc;
    HIPOK(hipMemcpy(d_in.p, h_in.data(), in_total, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(d_jobs.p, jobs.data(), sizeof(lzf_compress_job) * n_jobs, hipMemcpyHostToDevice));
    if (!indep) {
        std::vector<lzf_u32_table> h_tabs(n_linked, tmpl);                      // :213-214 table = template.clone()
        HIPOK(hipMemcpy(d_tabs.p, h_tabs.data(), sizeof(lzf_u32_table) * n_linked, hipMemcpyHostToDevice));
        rc = d_tabptr.alloc(sizeof(void*) * n_jobs); if (rc != LZF_OK) return rc;
        rc = d_adds.alloc(sizeof(uint64_t) * n_jobs); if (rc != LZF_OK) return rc;
        HIPOK(hipMemcpy(d_tabptr.p, h_tabptr.data(), sizeof(void*) * n_jobs, hipMemcpyHostToDevice));
        HIPOK(hipMemcpy(d_adds.p, h_adds.data(), sizeof(uint64_t) * n_jobs, hipMemcpyHostToDevice));
    }
    // ---- launches: no host round trip between the steps
    for (size_t k = 0; k + 1 < step_off.size(); ++k) {
        const size_t a = step_off[k], cnt = step_off[k + 1] - a;
        if (!cnt) continue;
        if (!indep && k > 0) { rc = lzf_table_offset_batch(d_tabptr.as<void*>() + a, d_adds.as<uint64_t>() + a, (uint32_t)cnt, LZF_TABLE_U32, nullptr); if (rc != LZF_OK) return rc; }
        rc = lzf_compress_batch(d_jobs.as<lzf_compress_job>() + a, d_res.as<lzf_job_result>() + a, (uint32_t)cnt, LZF_KINDS_U32, nullptr);
        if (rc != LZF_OK) return rc;
    }
    HIPOK(hipDeviceSynchronize());
    std::vector<lzf_job_result> res(n_jobs);
    std::vector<uint8_t> h_out(out_total ? out_total : 1);
    HIPOK(hipMemcpy(res.data(), d_res.p, sizeof(lzf_job_result) * n_jobs, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(h_out.data(), d_out.p, out_total, hipMemcpyDeviceToHost));
    // ---- assemble every frame (:244-263, :277-281)
    std::vector<std::vector<size_t>> frame_jobs(n_frames);
    for (size_t q = 0; q < n_jobs; ++q) { auto& v = frame_jobs[job_frame[q]]; if (v.size() <= job_block[q]) v.resize(job_block[q] + 1); v[job_block[q]] = q; }
    for (uint32_t f = 0; f < n_frames; ++f) {
        if (status[f] != LZF_OK) continue;
        const size_t nb = fr[f].nb;
        std::vector<const uint8_t*> payload(nb ? nb : 1);
        std::vector<uint32_t> clen(nb ? nb : 1), rlen(nb ? nb : 1);
        int st = LZF_OK;
        for (size_t i = 0; i < nb && st == LZF_OK; ++i) {
            const size_t q = frame_jobs[f][i], off = i * bs, n = in_len[f] - off < bs ? in_len[f] - off : bs;
            rlen[i] = (uint32_t)n;
            if (res[q].status == LZF_OK) { clen[i] = (uint32_t)res[q].out_len; payload[i] = h_out.data() + job_out_off[q]; }
            else if (res[q].status == LZF_OUTPUT_FULL) { clen[i] = UINT32_MAX; payload[i] = in[f] + off; }   // :250-255
            else st = res[q].status;
        }
        if (st != LZF_OK) { status[f] = st; continue; }
        uint32_t content = 0;
        if (s->content_checksum) content = lzf_xxh32(in[f], in_len[f], 0);    // :233-235
        status[f] = lzf_frame_assemble(s, (uint32_t)nb, payload.data(), clen.data(), rlen.data(), content, out[f], out_cap[f], &out_len[f]);
    }
    return LZF_OK;
}


int lzf_frame_decompress_many(uint32_t n_frames, const uint8_t* const* in, const size_t* in_len, const uint8_t* dict, size_t dict_len,
                              uint8_t* const* out, const size_t* out_cap, size_t* out_len, size_t* consumed, int* status) {
    if (n_frames && (!in || !in_len || !out || !out_cap || !out_len || !status)) return LZF_E_INVALID;
    if (!dict) dict_len = 0;
    struct Fr {
        lzf_frame_info fi; std::vector<Blk> blocks; FrameScan sc; bool live = false, linked = false;
        size_t in_off = 0, out_off = 0, out_size = 0;      // device offsets: the frame's bytes; linked: the stream's output buffer
        std::vector<size_t> job;                             // per block: job index or SIZE_MAX (stored)
        std::vector<size_t> slot;                            // independent: device offset of the block's output slot
        uint32_t chain = 0;                                  // linked: index among the linked streams
    };
    std::vector<Fr> fr(n_frames);
    size_t in_total = 0, out_total = 0, n_jobs = 0, max_steps = 0;
    uint32_t n_chain = 0;
    for (uint32_t f = 0; f < n_frames; ++f) {
        Fr& F = fr[f];
        out_len[f] = 0; if (consumed) consumed[f] = 0;
        const int rc = lzf_frame_read_header(in[f], in_len[f], &F.fi);
        if (rc != LZF_OK) { status[f] = rc; if (consumed) consumed[f] = rc == LZF_F_INPUT_ERROR ? in_len[f] : 0; continue; }
        status[f] = LZF_OK; F.live = true; F.linked = !(F.fi.flags & FL_INDEP);
        scan_blocks(in[f], in_len[f], F.fi, F.blocks, F.sc);
        if (consumed) consumed[f] = F.sc.consumed;
        const size_t nb = F.blocks.size(), bmax = (size_t)F.fi.block_maxsize;
        F.in_off = in_total; in_total = up256(in_total + in_len[f]);
        F.job.assign(nb, SIZE_MAX); F.slot.assign(nb, 0);
        size_t sumbl = 0;
        for (size_t i = 0; i < nb; ++i) if (F.blocks[i].compressed) { F.job[i] = 0; sumbl += F.blocks[i].len; }
        if (F.linked) {
            // a block may run past its limit by its literals before the stream is stopped: room for that
            if (nb) { F.chain = n_chain++; F.out_off = out_total; F.out_size = nb * bmax + sumbl + 64; out_total = up256(out_total + F.out_size); if (nb > max_steps) max_steps = nb; }
        } else {
            for (size_t i = 0; i < nb; ++i) if (F.blocks[i].compressed) { F.slot[i] = out_total; out_total = up256(out_total + bmax + F.blocks[i].len); }   // limit + C
        }
    }
    // ---- job list ordered by step: step 0 = every block of the independent frames + block 0 of the linked streams
    std::vector<lzf_decompress_job> jobs;
    std::vector<size_t> step_off;
    DBuf d_in, d_out, d_dict, d_jobs, d_res, d_steps, d_state;
    int rc = d_in.alloc(in_total); if (rc != LZF_OK) return rc;
    rc = d_out.alloc(out_total); if (rc != LZF_OK) return rc;
    if (dict_len) { rc = d_dict.alloc(dict_len); if (rc != LZF_OK) return rc; HIPOK(hipMemcpy(d_dict.p, dict, dict_len, hipMemcpyHostToDevice)); }
    uint8_t* const din = d_in.as<uint8_t>(); uint8_t* const dout = d_out.as<uint8_t>();
    const size_t n_steps = max_steps > 1 ? max_steps : 1;
    std::vector<lzf_chain_step> csteps((size_t)n_chain * n_steps);
    for (size_t k = 0; k < n_steps; ++k) {
        step_off.push_back(jobs.size());
        for (uint32_t f = 0; f < n_frames; ++f) {
            Fr& F = fr[f];
            if (!F.live) continue;
            const size_t nb = F.blocks.size(), bmax = (size_t)F.fi.block_maxsize;
            auto add_job = [&](size_t i) {
                lzf_decompress_job j;
                memset(&j, 0, sizeof j);
                j.input = din + F.in_off + (F.blocks[i].data - in[f]); j.input_len = F.blocks[i].len;
                j.prefix = d_dict.as<uint8_t>(); j.prefix_len = dict_len;                       // :239-245
                if (F.linked) { j.out = dout + F.out_off; j.out_cap = bmax + F.blocks[i].len; j.output_limit = bmax; }   // (patched per step)
                else { j.out = dout + F.slot[i]; j.out_cap = bmax + F.blocks[i].len; j.output_limit = bmax; }           // :248
                F.job[i] = jobs.size(); jobs.push_back(j);
            };
            if (!F.linked) { if (k == 0) for (size_t i = 0; i < nb; ++i) if (F.blocks[i].compressed) add_job(i); continue; }
            if (!nb) continue;
            lzf_chain_step& cs = csteps[k * n_chain + F.chain];
            memset(&cs, 0, sizeof cs);
            cs.prev_job = (k > 0 && k - 1 < nb && F.blocks[k - 1].compressed) ? (uint32_t)F.job[k - 1] : UINT32_MAX;
            cs.job = UINT32_MAX; cs.out = dout + F.out_off; cs.block_maxsize = bmax;
            if (k < nb) {
                if (F.blocks[k].compressed) { add_job(k); cs.job = (uint32_t)F.job[k]; }
                else { cs.stored_len = F.blocks[k].len; cs.stored_src = din + F.in_off + (F.blocks[k].data - in[f]); }
            }
        }
    }
    step_off.push_back(jobs.size());
    n_jobs = jobs.size();
    std::vector<lzf_job_result> res(n_jobs ? n_jobs : 1);
    if (in_total) for (uint32_t f = 0; f < n_frames; ++f) if (fr[f].live && in_len[f]) HIPOK(hipMemcpy(din + fr[f].in_off, in[f], in_len[f], hipMemcpyHostToDevice));
    if (n_jobs || n_chain) {
        rc = d_jobs.alloc(sizeof(lzf_decompress_job) * n_jobs); if (rc != LZF_OK) return rc;
        rc = d_res.alloc(sizeof(lzf_job_result) * n_jobs); if (rc != LZF_OK) return rc;
        if (n_jobs) HIPOK(hipMemcpy(d_jobs.p, jobs.data(), sizeof(lzf_decompress_job) * n_jobs, hipMemcpyHostToDevice));
        if (n_chain) {
            rc = d_steps.alloc(sizeof(lzf_chain_step) * csteps.size()); if (rc != LZF_OK) return rc;
            rc = d_state.alloc(sizeof(lzf_chain_state) * n_chain); if (rc != LZF_OK) return rc;
            HIPOK(hipMemcpy(d_steps.p, csteps.data(), sizeof(lzf_chain_step) * csteps.size(), hipMemcpyHostToDevice));
            HIPOK(hipMemset(d_state.p, 0, sizeof(lzf_chain_state) * n_chain));
        }
        for (size_t k = 0; k < n_steps; ++k) {
            if (n_chain) { rc = lzf_chain_decompress_step(d_steps.as<lzf_chain_step>() + k * n_chain, d_state.as<lzf_chain_state>(), n_chain, d_jobs.as<lzf_decompress_job>(), d_res.as<lzf_job_result>(), nullptr); if (rc != LZF_OK) return rc; }
            const size_t a = step_off[k], cnt = step_off[k + 1] - a;
            if (cnt) { rc = lzf_decompress_batch(d_jobs.as<lzf_decompress_job>() + a, d_res.as<lzf_job_result>() + a, (uint32_t)cnt, nullptr); if (rc != LZF_OK) return rc; }
        }
        HIPOK(hipDeviceSynchronize());
        if (n_jobs) HIPOK(hipMemcpy(res.data(), d_res.p, sizeof(lzf_job_result) * n_jobs, hipMemcpyDeviceToHost));
    }
    // ---- per frame: the delivery loop of lzf_frame_decompress, on the block lengths the device reports
    for (uint32_t f = 0; f < n_frames; ++f) {
        Fr& F = fr[f];
        if (!F.live) continue;
        const size_t nb = F.blocks.size(), bmax = (size_t)F.fi.block_maxsize;
        const bool csum = F.fi.flags & FL_CSUM;
        size_t w = 0, hist = 0;        // bytes delivered; linked: length of the stream's device buffer so far
        int st = LZF_OK; bool stopped = false;
        for (size_t i = 0; i < nb; ++i) {
            size_t n; const uint8_t* dsrc;                                                    // device address of the block's bytes
            if (F.blocks[i].compressed) {
                const lzf_job_result& r = res[F.job[i]];
                if (r.status != LZF_OK) { st = r.status; break; }                             // CodecError
                if (F.linked) { n = (size_t)r.out_len - hist; dsrc = dout + F.out_off + hist; }
                else { n = (size_t)r.out_len; dsrc = dout + F.slot[i]; }
            } else { n = F.blocks[i].len; dsrc = nullptr; }                                    // :250 stored: the bytes are in `in`
            hist += n;
            if (n > bmax) { st = LZF_F_BLOCK_SIZE_OVERFLOW; break; }                          // :272-274
            if (out_cap[f] - w < n) { st = LZF_OUT_CAPACITY; break; }
            if (n && dsrc) HIPOK(hipMemcpy(out[f] + w, dsrc, n, hipMemcpyDeviceToHost));
            else if (n) memcpy(out[f] + w, F.blocks[i].data, n);
            w += n;
            if (n == 0) { stopped = true; break; }                                            // the io::Read adapter stops at an empty block (:52-71,:286)
        }
        out_len[f] = w;
        if (st != LZF_OK) { status[f] = st; continue; }
        if (stopped) continue;
        if (F.sc.err != LZF_OK) { status[f] = F.sc.err; continue; }
        if (F.sc.endmark && csum && F.sc.want_content != lzf_xxh32(out[f], w, 0)) status[f] = LZF_F_FRAME_CHECKSUM_FAIL;   // :207-211
    }
    return LZF_OK;
}

}  // extern "C"
