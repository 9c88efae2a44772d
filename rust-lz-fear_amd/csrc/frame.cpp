// frame.cpp — host-side LZ4 frame layer over the GPU block codec (include/lzfear_frame.h).
//
// Mirrors lz-fear's src/framed/{compress,decompress,header}.rs; every block goes through the HIP
// kernels via lzf_compress_batch_host / lzf_decompress_batch_host — there is no CPU codec here.
// Independent-block frames submit all their blocks as ONE batch; linked-block frames are
// sequential by construction (table and 64 KiB window carry, compress.rs:271-275,
// decompress.rs:253-269) and run one block per call.
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

// compress.rs:160-282
int lzf_frame_compress(const lzf_settings* s, const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, size_t* out_len) {
    *out_len = 0;
    uint8_t bd;
    int rc = bd_new(s->block_size, &bd);                                       // :183
    if (rc != LZF_OK) return rc;
    if (out_cap < lzf_frame_compress_bound(s, in_len)) return LZF_OUT_CAPACITY;
    const size_t bs = (size_t)s->block_size;
    const uint8_t* dict = s->dictionary;
    const size_t dict_len = dict ? (size_t)s->dictionary_len : 0;
    const size_t n_blocks = (in_len + bs - 1) / bs;
    std::vector<lzf_job_result> res(n_blocks ? n_blocks : 1);
    std::vector<std::vector<uint8_t>> comp(n_blocks);
    lzf_u32_table tmpl;
    if (dict_len >= 8) { rc = seeded_template(dict, dict_len, &tmpl); if (rc != LZF_OK) return rc; }

    if (s->independent_blocks) {
        // every block is a self-contained job (prefix = dictionary, template table cloned read-only)
        std::vector<lzf_compress_job> jobs(n_blocks);
        std::vector<std::vector<uint8_t>> inbuf(dict_len ? n_blocks : 0);
        std::vector<lzf_u32_table> tabs(dict_len >= 8 ? n_blocks : 0);
        for (size_t i = 0; i < n_blocks; ++i) {
            const size_t off = i * bs, n = in_len - off < bs ? in_len - off : bs;
            comp[i].resize(n);
            lzf_compress_job& j = jobs[i];
            memset(&j, 0, sizeof j);
            if (dict_len) {                                                    // :218,:268 in_buffer = dict ++ block
                inbuf[i].resize(dict_len + n);
                memcpy(inbuf[i].data(), dict, dict_len);
                memcpy(inbuf[i].data() + dict_len, in + off, n);
                j.input = inbuf[i].data(); j.input_len = dict_len + n; j.cursor = dict_len;
            } else { j.input = in + off; j.input_len = n; j.cursor = 0; }
            j.out = comp[i].data(); j.out_cap = n;                             // :242 cap = read_bytes
            j.table_kind = LZF_TABLE_U32;                                      // :202
            if (dict_len >= 8) { tabs[i] = tmpl; j.table = &tabs[i]; j.flags = LZF_CJOB_TABLE_READONLY; }   // :220,:270
        }
        rc = lzf_compress_batch_host(jobs.data(), res.data(), (uint32_t)n_blocks);
        if (rc != LZF_OK) return rc;
    } else {
        // linked blocks: table + last 64 KiB carried (:271-275)
        lzf_u32_table table;
        if (dict_len >= 8) table = tmpl; else memset(&table, 0, sizeof table);
        std::vector<uint8_t> in_buffer(dict, dict + dict_len);
        for (size_t i = 0; i < n_blocks; ++i) {
            const size_t off = i * bs, n = in_len - off < bs ? in_len - off : bs;
            const size_t window_offset = in_buffer.size();                     // :222
            in_buffer.insert(in_buffer.end(), in + off, in + off + n);
            comp[i].resize(n);
            lzf_compress_job j;
            memset(&j, 0, sizeof j);
            j.input = in_buffer.data(); j.input_len = in_buffer.size(); j.cursor = window_offset;
            j.out = comp[i].data(); j.out_cap = n; j.table = &table; j.table_kind = LZF_TABLE_U32;
            rc = lzf_compress_batch_host(&j, &res[i], 1);
            if (rc != LZF_OK) return rc;
            if (in_buffer.size() > LZF_WINDOW_SIZE) {
                const size_t forget = in_buffer.size() - LZF_WINDOW_SIZE;
                table.offset += forget;                                        // mod.rs:72-74
                in_buffer.erase(in_buffer.begin(), in_buffer.begin() + (ptrdiff_t)forget);
            }
        }
    }
    // assemble (:244-263, :277-281)
    std::vector<const uint8_t*> payload(n_blocks);
    std::vector<uint32_t> clen(n_blocks), rlen(n_blocks);
    for (size_t i = 0; i < n_blocks; ++i) {
        const size_t off = i * bs, n = in_len - off < bs ? in_len - off : bs;
        rlen[i] = (uint32_t)n;
        if (res[i].status == LZF_OK) { clen[i] = (uint32_t)res[i].out_len; payload[i] = comp[i].data(); }
        else if (res[i].status == LZF_OUTPUT_FULL) { clen[i] = UINT32_MAX; payload[i] = in + off; }   // :250-255
        else return res[i].status;
    }
    uint32_t content = 0;
    if (s->content_checksum) content = lzf_xxh32(in, in_len, 0);               // :233-235
    return lzf_frame_assemble(s, (uint32_t)n_blocks, payload.data(), clen.data(), rlen.data(), content, out, out_cap, out_len);
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

// decompress.rs:198-288
int lzf_frame_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len,
                         uint8_t* out, size_t out_cap, size_t* out_len, size_t* consumed) {
    *out_len = 0; if (consumed) *consumed = 0;
    lzf_frame_info fi;
    int rc = lzf_frame_read_header(in, in_len, &fi);
    if (rc != LZF_OK) { if (consumed) *consumed = rc == LZF_F_INPUT_ERROR ? in_len : 0; return rc; }
    const size_t bmax = (size_t)fi.block_maxsize;
    const bool linked = !(fi.flags & FL_INDEP), bsum = fi.flags & FL_BLOCKSUM, csum = fi.flags & FL_CSUM;

    // ---- scan the block structure (u32 length hops, :205-235)
    struct Blk { const uint8_t* data; uint32_t len; bool compressed; };
    std::vector<Blk> blocks;
    size_t r = fi.header_len;
    int scan_err = LZF_OK;        // structural error that ends the scan (reported in stream order)
    bool endmark = false; uint32_t want_content = 0;
    for (;;) {
        if (in_len - r < 4) { scan_err = LZF_F_INPUT_ERROR; r = in_len; break; }
        uint32_t bl = rd32(in + r); r += 4;
        if (bl == 0) {                                                          // :206-215
            if (csum) { if (in_len - r < 4) { scan_err = LZF_F_INPUT_ERROR; r = in_len; break; } want_content = rd32(in + r); r += 4; }
            endmark = true; break;
        }
        const bool compressed = (bl & INCOMPRESSIBLE) == 0; bl &= ~INCOMPRESSIBLE;
        if (bl > (uint32_t)bmax) { scan_err = LZF_F_BLOCK_SIZE_OVERFLOW; break; }             // :220-222
        if (in_len - r < bl) { scan_err = LZF_F_INPUT_ERROR; r = in_len; break; }             // :226
        const uint8_t* data = in + r; r += bl;
        if (bsum) {                                                                           // :228-235
            if (in_len - r < 4) { scan_err = LZF_F_INPUT_ERROR; r = in_len; break; }
            const uint32_t c = rd32(in + r); r += 4;
            if (c != lzf_xxh32(data, bl, 0)) { scan_err = LZF_F_BLOCK_CHECKSUM_FAIL; break; }
        }
        blocks.push_back({data, bl, compressed});
    }
    if (consumed) *consumed = r;

    // ---- decode
    const size_t nb = blocks.size();
    std::vector<lzf_job_result> res(nb ? nb : 1);
    std::vector<std::vector<uint8_t>> dec(nb);
    Xxh32 content;
    size_t w = 0;
    int status = LZF_OK;
    bool stopped = false;         // the io::Read adapter stops at a block that yields 0 bytes (:52-71,:286)
    auto deliver = [&](size_t i, const uint8_t* p, size_t n) -> bool {
        if (n > bmax) { status = LZF_F_BLOCK_SIZE_OVERFLOW; return false; }                   // :272-274
        if (out_cap - w < n) { status = LZF_OUT_CAPACITY; return false; }
        memcpy(out + w, p, n); w += n;
        if (csum) content.update(p, n);                                                       // :276-278
        if (n == 0) { stopped = true; return false; }
        (void)i; return true;
    };
    if (!linked) {
        std::vector<lzf_decompress_job> jobs; std::vector<size_t> jidx;
        for (size_t i = 0; i < nb; ++i) {
            if (!blocks[i].compressed) continue;
            dec[i].resize(bmax + blocks[i].len);            // limit + C: exact malformed-input parity
            lzf_decompress_job j;
            memset(&j, 0, sizeof j);
            j.input = blocks[i].data; j.input_len = blocks[i].len;
            j.prefix = dict; j.prefix_len = dict_len;                                         // :244
            j.out = dec[i].data(); j.out_cap = dec[i].size(); j.output_limit = bmax;          // :248
            jobs.push_back(j); jidx.push_back(i);
        }
        std::vector<lzf_job_result> jr(jobs.size() ? jobs.size() : 1);
        if (!jobs.empty()) { rc = lzf_decompress_batch_host(jobs.data(), jr.data(), (uint32_t)jobs.size()); if (rc != LZF_OK) return rc; }
        for (size_t k = 0; k < jobs.size(); ++k) res[jidx[k]] = jr[k];
        for (size_t i = 0; i < nb; ++i) {
            if (blocks[i].compressed) {
                if (res[i].status != LZF_OK) { status = res[i].status; break; }               // CodecError
                if (!deliver(i, dec[i].data(), (size_t)res[i].out_len)) break;
            } else if (!deliver(i, blocks[i].data, blocks[i].len)) break;                     // :250
        }
    } else {
        std::vector<uint8_t> window;                                                          // carryover_window :144-148
        std::vector<uint8_t> buf(bmax * 2 + 16);
        for (size_t i = 0; i < nb; ++i) {
            if (window.empty()) window.assign(dict, dict + dict_len);                         // :239-241
            size_t n = 0; const uint8_t* p;
            if (blocks[i].compressed) {
                if (buf.size() < bmax + blocks[i].len) buf.resize(bmax + blocks[i].len);
                lzf_decompress_job j;
                memset(&j, 0, sizeof j);
                j.input = blocks[i].data; j.input_len = blocks[i].len;
                j.prefix = window.data(); j.prefix_len = window.size();
                j.out = buf.data(); j.out_cap = bmax + blocks[i].len; j.output_limit = bmax;
                lzf_job_result jr;
                rc = lzf_decompress_batch_host(&j, &jr, 1);
                if (rc != LZF_OK) return rc;
                if (jr.status != LZF_OK) { status = jr.status; break; }
                n = (size_t)jr.out_len; p = buf.data();
            } else { n = blocks[i].len; p = blocks[i].data; }
            // window update :253-269 (before the size check, like the reference)
            if (n < LZF_WINDOW_SIZE) {
                const size_t avail = window.size() + n;
                if (avail >= LZF_WINDOW_SIZE) window.erase(window.begin(), window.begin() + (ptrdiff_t)(avail - LZF_WINDOW_SIZE));
                window.insert(window.end(), p, p + n);
            } else window.assign(p + n - LZF_WINDOW_SIZE, p + n);
            if (!deliver(i, p, n)) break;
        }
    }
    *out_len = w;
    if (status != LZF_OK) return status;
    if (stopped) return LZF_OK;
    if (scan_err != LZF_OK) return scan_err;
    if (endmark && csum && want_content != content.digest()) return LZF_F_FRAME_CHECKSUM_FAIL;    // :207-211
    return LZF_OK;
}

}  // extern "C"
